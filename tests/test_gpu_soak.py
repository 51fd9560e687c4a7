"""GPU: a short soak of every kernel shape whose waves hand data to each other through LDS behind flag synchronisation (pair_sync /
pair_arrive + pair_wait: arrival counters, no fence -- the ordering rests on DS instructions of a wave executing in order).  The same
batch launched again and again must give the same words every time; a rare ordering bug (or a compiler upgrade that moves an access
across a hand-off) shows as a rare differing launch.  scripts/soak.py is the long form (5,000 launches; profiles/r0*/soak_*.log)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_repeated_launches_of_every_flag_synchronised_shape_give_the_same_words():
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import soak
    lines = []
    # N = 1024: 4 gates per workgroup, 3 and 2, a ragged batch, five and six gates per CU time-sliced (k_bootstrap_pair_rr: a gate changes hands
    # between wave pairs through LDS flags); N = 2048: the ping-pong trade at 4 / 3 gates
    # per workgroup, the duplicated first stage at 2; the NTT backend's pair-synchronised shapes at a quarter of the launches
    shapes = ((1024, ("fft", "ntt"), (1024, 768, 512, 300, 1280, 1500)), (2048, ("fft", "ntt"), (1024, 768, 512)))
    bad = soak.run(60, shapes, emit=lines.append)
    assert bad == 0, "\n".join(lines)
    assert len(lines) == 18
