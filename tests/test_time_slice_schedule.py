"""CPU: the item schedule of k_bootstrap_pair_rr (rustfhe_amd/csrc/rtfhe_kernels_pair_rr.hpp), restated and checked for what the kernel relies on.

A workgroup's four wave pairs ("slots") work through the CMUX steps of its gc gates as items t = step * gc + gate; slot s takes t = s, s + 4, ...
and walks them with the kernel's own update (gate += 4; on reaching gc: gate -= gc, step += 1).  Item (gate, step) may start once the same
gate's step - 1 is done -- by whichever slot ran it.  Checked here: the walk visits exactly the slot's items in order; every item is visited once;
a gate's previous step always lies earlier in the global order, in another slot when gc > 4, so the waits cannot form a cycle; and a
free-running simulation with arbitrary item durations finishes every item, in about gc / 4 rounds when the durations are equal."""
import random

SLOTS = 4


def walk(slot, gc, steps):
    """the kernel's loop: (gate, step) items of one slot, in program order"""
    gl, i, out = slot, 0, []
    while i < steps:
        out.append((gl, i))
        gl += SLOTS
        if gl >= gc:
            gl -= gc
            i += 1
    return out


def test_the_walk_visits_the_slots_items_of_the_round_robin_order():
    for gc in (4, 5, 6, 7, 8):
        for steps in (0, 1, 2, 7, 635):
            seen = {}
            for s in range(SLOTS):
                items = walk(s, gc, steps)
                ts = [i * gc + g for g, i in items]
                assert ts == list(range(s, gc * steps, SLOTS)), (gc, steps, s)
                for k, it in enumerate(items):
                    assert it not in seen
                    seen[it] = (s, k)
            assert len(seen) == gc * steps
            for (g, i), (s, k) in seen.items():
                if i == 0:
                    continue
                ps, pk = seen[(g, i - 1)]
                assert (i - 1) * gc + g < i * gc + g
                if gc > SLOTS:
                    assert ps != s or pk < k          # the previous step ran on another slot, or earlier on this one
                else:
                    assert ps == s and pk == k - 1    # four gates: every gate stays on its slot (k_bootstrap_pair's assignment)


def simulate(gc, steps, duration):
    """free-running slots: an item starts when its slot is free AND the gate's previous step is done; returns the finish time of everything"""
    seqs = [walk(s, gc, steps) for s in range(SLOTS)]
    pos, free_at, done_at = [0] * SLOTS, [0.0] * SLOTS, {}
    remaining = sum(len(q) for q in seqs)
    while remaining:
        progressed = False
        for s in range(SLOTS):
            if pos[s] == len(seqs[s]):
                continue
            g, i = seqs[s][pos[s]]
            if i > 0 and (g, i - 1) not in done_at:
                continue                              # the slot waits at its flag
            start = max(free_at[s], done_at.get((g, i - 1), 0.0))
            done_at[(g, i)] = free_at[s] = start + duration(g, i, s)
            pos[s] += 1
            remaining -= 1
            progressed = True
        assert progressed, "deadlock"
    return max(done_at.values()) if done_at else 0.0


def test_free_running_slots_never_deadlock_and_take_gc_over_four_rounds():
    rng = random.Random(7)
    for gc in (4, 5, 6):
        t = simulate(gc, 635, lambda g, i, s: 1.0)
        assert abs(t - gc * 635 / SLOTS) <= 2.0, (gc, t)          # equal items: gc / 4 rounds (a round = 635 items), up to the ramp at both ends
        for _ in range(5):                                        # uneven items (a slow pair, jitter): everything still finishes
            slow = rng.randrange(SLOTS)
            simulate(gc, 97, lambda g, i, s: rng.uniform(0.5, 1.5) * (3.0 if s == slow else 1.0))
