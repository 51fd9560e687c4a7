"""Gate-netlist front-end and dependency-wave scheduler (BASELINE config 4).

Build-side counterpart of the reference's `nander` crate: same gate vocabulary and expression grammar
(nander/src/lib.rs:19-38 `Logip`, :64-89 `LogicExpr` + eager tree evaluator, :90-172 recursive-descent parser
over the tokens `0 1 & | ^ ! $ ( )`, left-associative, no precedence).  Where the reference walks the tree and
bootstraps one gate at a time, this levelises a netlist and runs every dependency wave as ONE launch of the
bootstrap kernel (`rtfhe_circuit_wave_dev`: per-gate opcode + wire indices), optionally for many independent
instances of the circuit at once ("replicas") so that waves are wide enough to fill the GPU.
"""
import numpy as np

from ._ffi import AND, ANDNY, COPY, NAND, NOT, OR, XOR

_NAMES = {NAND: "nand", AND: "and", OR: "or", XOR: "xor", NOT: "not", COPY: "copy", ANDNY: "andny"}


def plain_gate(op, a, b):
    """Truth table of one opcode on plain bits (used to check netlists without any ciphertext)."""
    if op == NAND:
        return 1 - (a & b)
    if op == AND:
        return a & b
    if op == OR:
        return a | b
    if op == XOR:
        return a ^ b
    if op == NOT:
        return 1 - a
    if op == ANDNY:
        return (1 - a) & b
    return a


class Netlist:
    """Wires are integers.  Wires 0 and 1 are the constants false/true (trivial ciphertexts, AsLogic,
    hom_nand/src/tlwe.rs:80-87); inputs follow; every gate defines a new wire."""

    FALSE, TRUE = 0, 1

    def __init__(self):
        self.num_inputs = 0
        self.gates = []          # (op, a, b) defining wire 2 + num_inputs + index
        self.outputs = []
        self._frozen_inputs = False

    # ---- construction ----
    def input(self):
        assert not self._frozen_inputs, "declare all inputs before the first gate"
        self.num_inputs += 1
        return 1 + self.num_inputs

    def inputs(self, k):
        return [self.input() for _ in range(k)]

    def gate(self, op, a, b=None):
        self._frozen_inputs = True
        b = a if b is None else b
        w = 2 + self.num_inputs + len(self.gates)
        assert 0 <= a < w and 0 <= b < w
        self.gates.append((op, a, b))
        return w

    # Logip vocabulary (nander/src/lib.rs:19-62: TFHE implements each with its own bootstrap)
    def nand(self, a, b): return self.gate(NAND, a, b)
    def and_(self, a, b): return self.gate(AND, a, b)
    def or_(self, a, b): return self.gate(OR, a, b)
    def xor(self, a, b): return self.gate(XOR, a, b)
    def not_(self, a): return self.gate(NOT, a)

    def mux(self, c, in0, in1):
        """(in1 & c) | (in0 & !c), hom_mux (hom_nand/src/tfhe.rs:27-40): two ANDs + one bootstrap = 2 levels."""
        return self.gate(OR, self.gate(AND, c, in1), self.gate(ANDNY, c, in0))

    def output(self, w):
        self.outputs.append(w)
        return w

    @property
    def num_wires(self):
        return 2 + self.num_inputs + len(self.gates)

    # ---- scheduling ----
    def levels(self):
        """Topological levelisation: gate level = 1 + max(level of its operands); inputs/constants are level 0.
        Returns a list of waves, each a list of gate indices."""
        base = 2 + self.num_inputs
        lvl = [0] * self.num_wires
        waves = []
        for gi, (op, a, b) in enumerate(self.gates):
            l = 1 + max(lvl[a], lvl[b])
            lvl[base + gi] = l
            while len(waves) < l:
                waves.append([])
            waves[l - 1].append(gi)
        return waves

    def evaluate_plain(self, bits):
        """Reference semantics on plain bits."""
        assert len(bits) == self.num_inputs
        v = [0, 1] + [int(x) & 1 for x in bits] + [0] * len(self.gates)
        base = 2 + self.num_inputs
        for gi, (op, a, b) in enumerate(self.gates):
            v[base + gi] = plain_gate(op, v[a], v[b])
        return [v[w] for w in self.outputs]

    def describe(self):
        w = self.levels()
        return {"inputs": self.num_inputs, "gates": len(self.gates), "depth": len(w), "wave_sizes": [len(x) for x in w]}


# ---- the reference's expression grammar (nander/src/lib.rs:90-172) ----------------------------------

def parse_logic_expr(text, net=None):
    """Parses a constant expression over 0/1 with & | ^ $ (binary, left-assoc, no precedence), ! (prefix) and
    parentheses into a Netlist; returns (netlist, output wire).  Errors mirror the reference's messages."""
    net = net or Netlist()
    s = "".join(text.split())
    pos = [0]

    def peek():
        return s[pos[0]] if pos[0] < len(s) else None

    def take():
        c = peek()
        pos[0] += 1
        return c

    def elem():
        c = take()
        if c is None:
            raise ValueError("invalid element. this is none")
        if c == "0":
            return Netlist.FALSE
        if c == "1":
            return Netlist.TRUE
        if c == "(":
            e = binary()
            if take() != ")":
                raise ValueError("braket is not closed")
            return e
        raise ValueError("invalid element")

    def mono():
        if peek() == "!":
            take()
            return net.not_(mono())
        return elem()

    def binary():
        lhs = mono()
        while True:
            c = peek()
            if c == "&":
                take(); lhs = net.and_(lhs, mono())
            elif c == "|":
                take(); lhs = net.or_(lhs, mono())
            elif c == "^":
                take(); lhs = net.xor(lhs, mono())
            elif c == "$":
                take(); lhs = net.nand(lhs, mono())
            else:
                return lhs

    out = binary()
    net.output(out)
    return net, out


# ---- circuits ---------------------------------------------------------------------------------------

def full_adder_nand(net, a, b, cin):
    """9-NAND full adder: sum = a ^ b ^ cin, carry = majority."""
    n1 = net.nand(a, b)
    n2 = net.nand(a, n1)
    n3 = net.nand(b, n1)
    x = net.nand(n2, n3)          # a ^ b
    n5 = net.nand(x, cin)
    n6 = net.nand(x, n5)
    n7 = net.nand(cin, n5)
    s = net.nand(n6, n7)          # a ^ b ^ cin
    c = net.nand(n5, n1)          # (a & b) | (cin & (a ^ b))
    return s, c


def full_adder_mixed(net, a, b, cin):
    """5-gate full adder with the engine's native XOR/AND/OR bootstraps."""
    x = net.xor(a, b)
    s = net.xor(x, cin)
    c = net.or_(net.and_(a, b), net.and_(x, cin))
    return s, c


def ripple_carry_adder(nbits=8, nand_only=True):
    """nbits + nbits -> nbits + carry.  Inputs: a0..a{n-1}, b0..b{n-1} (LSB first); outputs: s0..s{n-1}, carry."""
    net = Netlist()
    a = net.inputs(nbits)
    b = net.inputs(nbits)
    fa = full_adder_nand if nand_only else full_adder_mixed
    c = Netlist.FALSE
    for i in range(nbits):
        s, c = fa(net, a[i], b[i], c)
        net.output(s)
    net.output(c)
    return net


def _prefix_adder_nand(nbits):
    """prefix_adder in NAND gates only (what the reference's nander evaluates).  Both polarities of a group generate are kept:
    G' = G | (P & Gl) = NAND(~G, NAND(P, Gl)) needs ~G one level later than P and Gl, which is when NAND(G, G) of the stage before
    delivers it -- so a prefix stage still costs two levels.  Propagate is a | b = NAND(~a, ~b) (one level earlier than a ^ b);
    the half sums a ^ b reuse ~g = NAND(a, b)."""
    net = Netlist()
    a = net.inputs(nbits)
    b = net.inputs(nbits)
    neg_memo = {}

    def neg(w):
        if w not in neg_memo:
            neg_memo[w] = net.nand(w, w)
        return neg_memo[w]
    gn = [net.nand(a[i], b[i]) for i in range(nbits)]
    x = [net.nand(net.nand(a[i], gn[i]), net.nand(b[i], gn[i])) for i in range(nbits)]      # a ^ b
    G = []
    for i in range(nbits):
        gi = net.nand(gn[i], gn[i])
        neg_memo[gi] = gn[i]
        G.append(gi)
    P = [net.nand(neg(a[i]), neg(b[i])) for i in range(nbits)]                             # a | b
    d = 1
    while d < nbits:
        last = 2 * d >= nbits
        newG, newP = list(G), list(P)
        for i in range(d, nbits):
            newG[i] = net.nand(neg(G[i]), net.nand(P[i], G[i - d]))
            if not last:
                pn = net.nand(P[i], P[i - d])
                newP[i] = net.nand(pn, pn)
        G, P = newG, newP
        d *= 2
    net.output(x[0])
    for i in range(1, nbits):
        c = G[i - 1]
        n1 = net.nand(x[i], c)
        net.output(net.nand(net.nand(x[i], n1), net.nand(c, n1)))
    net.output(G[nbits - 1])
    return net


def prefix_adder(nbits=8, nand_only=False):
    """The same nbits + nbits -> nbits + carry function as ripple_carry_adder, as a parallel-prefix (Kogge-Stone) netlist of the
    reference's AND / OR / XOR gates (hom_and / hom_or / hom_xor, hom_nand/src/tfhe.rs:27-71).  A dependency wave of up to a few hundred
    gates costs the engine what one gate costs, so only the DEPTH of a netlist matters for one addition: generate / propagate (one
    level), log2(nbits) prefix stages of two levels each -- (G, P) o (G', P') = (G | (P & G'), P & P') -- and the sum XORs: depth
    at most 2 + 2 log2(nbits); 7 for 8 bits (the low groups finish early) against 17 for the ripple-carry netlist of the same gates
    (more gates: 70 against 40).  nand_only=True: the same structure in NAND gates (depth 11 against 20 for the NAND ripple-carry)."""
    if nand_only:
        return _prefix_adder_nand(nbits)
    net = Netlist()
    a = net.inputs(nbits)
    b = net.inputs(nbits)
    p = [net.xor(a[i], b[i]) for i in range(nbits)]       # propagate (also the half sums)
    G = [net.and_(a[i], b[i]) for i in range(nbits)]      # generate
    P = list(p)
    d = 1
    while d < nbits:
        t = {i: net.and_(P[i], G[i - d]) for i in range(d, nbits)}
        last = 2 * d >= nbits                               # group propagates are not needed after the last stage
        P2 = {} if last else {i: net.and_(P[i], P[i - d]) for i in range(d, nbits)}
        G = [G[i] if i < d else net.or_(G[i], t[i]) for i in range(nbits)]
        P = [P[i] if i < d or last else P2[i] for i in range(nbits)]
        d *= 2
    net.output(p[0])
    for i in range(1, nbits):
        net.output(net.xor(p[i], G[i - 1]))                # G[i - 1] = carry into bit i
    net.output(G[nbits - 1])
    return net


# ---- execution on the engine ------------------------------------------------------------------------

class CircuitRunner:
    """Runs `replicas` independent instances of a netlist on one Engine; every dependency wave is one launch of
    replicas * wave_size gates.  Wire table: int32[replicas * num_wires][n + 1] resident in HBM."""

    def __init__(self, engine, net, replicas=1):
        import torch
        self.e, self.net, self.R = engine, net, replicas
        self.n1 = engine.p.n + 1
        W = net.num_wires
        base = 2 + net.num_inputs
        self.wires = torch.zeros((replicas * W, self.n1), dtype=torch.int32, device="cuda")
        # constants: trivial TLWE of -1/8 and +1/8 (b only)
        const = np.zeros((2, self.n1), np.uint32)
        const[0, -1], const[1, -1] = 0xE0000000, 0x20000000
        cz = torch.from_numpy(const.view(np.int32)).cuda()
        idx = torch.arange(replicas, device="cuda") * W
        self.wires[idx] = cz[0]
        self.wires[idx + 1] = cz[1]
        self.waves = []
        self._circuit, self._flat = None, None
        off = (np.arange(replicas, dtype=np.int64) * W)[:, None]
        for wave in net.levels():
            ops = np.array([net.gates[g][0] for g in wave], np.int32)
            i0 = np.array([net.gates[g][1] for g in wave], np.int64)
            i1 = np.array([net.gates[g][2] for g in wave], np.int64)
            io = np.array([base + g for g in wave], np.int64)
            pack = lambda x: torch.from_numpy((x[None, :] + off).reshape(-1).astype(np.int32)).cuda()
            self.waves.append((torch.from_numpy(np.tile(ops, replicas)).cuda(), pack(i0), pack(i1), pack(io), replicas * len(wave)))

    def set_inputs(self, cts):
        """cts: uint32[replicas][num_inputs][n+1] (numpy)."""
        import torch
        cts = np.ascontiguousarray(cts, np.uint32).reshape(self.R, self.net.num_inputs, self.n1)
        W = self.net.num_wires
        view = self.wires.view(self.R, W, self.n1)
        view[:, 2:2 + self.net.num_inputs] = torch.from_numpy(cts.view(np.int32)).cuda()

    def run(self, graph=True):
        """Evaluates the netlist.  graph=True (default): all dependency waves were recorded once into a HIP graph
        (rtfhe_circuit_create) and are replayed as ONE submission; graph=False: one rtfhe_circuit_wave_dev call per wave.
        Either way the call returns after the device has finished, and a gate the device skipped (wire index or opcode
        out of range) raises RtfheError HERE, not at some later unrelated sync."""
        import torch
        st = torch.cuda.current_stream().cuda_stream
        if graph and self.waves:
            if self._circuit is None:
                cat = lambda k: torch.cat([w[k] for w in self.waves]).contiguous()
                self._flat = (cat(0), cat(1), cat(2), cat(3))
                offs = np.concatenate([[0], np.cumsum([w[4] for w in self.waves])]).astype(np.int32)
                self._circuit = self.e.circuit_create(*self._flat, offs, self.wires, self.R * self.net.num_wires)
            self.e.circuit_launch(self._circuit, st)
        else:
            for ops, i0, i1, io, cnt in self.waves:
                self.e.circuit_wave_dev(ops, i0, i1, io, self.wires, self.R * self.net.num_wires, cnt, st)
        self.e.sync(st)
        return self

    def close(self):
        if getattr(self, "_circuit", None) is not None:
            self.e.circuit_destroy(self._circuit)
            self._circuit = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def outputs(self):
        """uint32[replicas][num_outputs][n+1]."""
        import torch
        W = self.net.num_wires
        view = self.wires.view(self.R, W, self.n1)
        idx = torch.tensor(self.net.outputs, device="cuda", dtype=torch.long)
        return view[:, idx].cpu().numpy().view(np.uint32)

    def wire(self, w):
        return self.wires.view(self.R, self.net.num_wires, self.n1)[:, w].cpu().numpy().view(np.uint32)


def eval_logic_expr(engine, text):
    """nander's REPL semantics (nander/src/main.rs:20-70): parse a constant expression, evaluate it homomorphically
    on trivial ciphertexts, return the output TLWE (uint32[n+1])."""
    net, _ = parse_logic_expr(text)
    r = CircuitRunner(engine, net, 1)
    if net.gates:
        r.run()
    return r.outputs()[0, 0]
