"""snf_edit_distance_batch (bit-parallel Myers, replaces edlib.align(a,b)['editDistance'] in SVGroup.align_call,
sv.py:280-289) against the exact two-row DP of the oracle.  edlib itself is absent: parity with edlib is the
mathematical definition of its default mode (global NW, unit costs) - "parity unpinned" against the library."""
import numpy as np
import pytest

from sniffles_amd import lib


def make_pairs(seed, sizes, alphabet=b"ACGT"):
    rng = np.random.default_rng(seed)
    pairs = [(b"", b""), (b"kitten", b"sitting"), (b"<DEL>", b"<DEL>"), (b"ACGT", b""), (b"", b"ACGTN"), (b"<INS>", b"<DEL>"),
             (b"A" * 64, b"A" * 64), (b"A" * 65, b"C" * 63), (b"ACGT" * 16, b"ACGT" * 16 + b"T")]
    al = list(alphabet)
    for n in sizes:
        a = bytes(rng.choice(al, n).astype(np.uint8))
        b = bytearray(a)
        for _ in range(int(rng.integers(0, max(2, n // 8)))):
            p = int(rng.integers(0, max(1, len(b))))
            op = rng.integers(0, 3)
            if op == 0 and b:
                b[p % len(b)] = int(rng.choice(al))
            elif op == 1:
                b.insert(p, int(rng.choice(al)))
            elif b:
                del b[p % len(b)]
        pairs.append((a, bytes(b)))
        if n > 4:
            pairs.append((a, bytes(rng.choice(al, int(rng.integers(1, n + 20))).astype(np.uint8))))  # unrelated
    return pairs


def test_edit_distance_emulated(oracle_mod):
    import emu.emu as E
    sizes = [1, 2, 63, 64, 65, 127, 128, 129, 300, 511, 513, 700] + list(np.random.default_rng(0).integers(1, 400, 60))
    pairs = make_pairs(1, sizes, alphabet=b"ACGTNacgt<>")
    got = lib.edit_distance_batch(pairs, _lib=E.lib())
    assert got.tolist() == [oracle_mod.edit_distance(a, b) for a, b in pairs]


def check_banded(pairs, exact, **kw):
    """edlib's `k`: the distance when it is <= k, otherwise -1 - for cut-offs below, at and above the true distance,
    a cut-off smaller than the length difference, 0, and cut-offs far beyond (no band)."""
    rng = np.random.default_rng(5)
    for mode in range(6):
        ks = []
        for (a, b), d in zip(pairs, exact):
            ks.append([d - 1, d, d + 1, abs(len(a) - len(b)) - 1, int(rng.integers(0, 2 * max(len(a), len(b)) + 2)), 0][mode])
        ks = [max(k, 0) if mode != 3 else k for k in ks]
        ks = [k if k >= 0 else 0 for k in ks]
        got = lib.edit_distance_batch(pairs, max_dist=ks, **kw).tolist()
        assert got == [d if d <= k else -1 for d, k in zip(exact, ks)], mode
    assert lib.edit_distance_batch(pairs, max_dist=-1, **kw).tolist() == exact


def test_edit_distance_banded_emulated(oracle_mod):
    import emu.emu as E
    sizes = [1, 2, 63, 64, 65, 127, 128, 129, 300, 511, 513, 700, 1500] + list(np.random.default_rng(7).integers(1, 400, 40))
    pairs = make_pairs(8, sizes, alphabet=b"ACGTN")
    check_banded(pairs, [oracle_mod.edit_distance(a, b) for a, b in pairs], _lib=E.lib())


@pytest.mark.gpu
def test_edit_distance_banded_gpu(oracle_mod):
    """thread form (<= 8 blocks, states in LDS), banded wave form (lane = block of the band, blocks rotating through the
    lanes beyond 4 kb) and the multi-pass wave form (bands wider than 63 blocks)."""
    sizes = [1, 63, 64, 65, 128, 300, 512, 513, 600, 1000, 2047, 4096, 4097, 5000, 6100, 9000, 12000] + \
        list(np.random.default_rng(4).integers(1, 3000, 80))
    pairs = make_pairs(6, sizes, alphabet=b"ACGTN")
    check_banded(pairs, [oracle_mod.edit_distance(a, b) for a, b in pairs])


@pytest.mark.gpu
def test_edit_distance_gpu_thread_and_wave_paths(oracle_mod):
    # <= 512: thread per pair; > 512: wave per pair; > 4096: several 64-block passes
    sizes = [1, 63, 64, 65, 128, 300, 512, 513, 600, 1000, 2047, 4096, 4097, 5000, 9000] + \
        list(np.random.default_rng(2).integers(1, 3000, 120))
    pairs = make_pairs(3, sizes, alphabet=b"ACGTN")
    got = lib.edit_distance_batch(pairs)
    assert got.tolist() == [oracle_mod.edit_distance(a, b) for a, b in pairs]


def test_population_variant_match_is_the_reference_rule(oracle_mod):
    """`snfp.PopulationVariant.match` (`/root/reference/src/sniffles/snfp.py:91-107`): position / length gate, then for insertions
    `edlib.align(self.alt, svcall.alt)['editDistance']` against `combine_pctseq` - restated with the exact DP (edlib is absent:
    parity unpinned against edlib itself) and compared with the batched GPU form."""
    import math
    import numpy as np
    import emu.emu as E
    from types import SimpleNamespace as NS
    from sniffles_amd import snfp
    from sniffles_amd.config import SnifflesConfig
    rng = np.random.default_rng(3)
    cfg = SnifflesConfig()

    def rnd(n):
        return "".join("ACGT"[i] for i in rng.integers(0, 4, n))
    pairs = []
    for _ in range(120):
        n = int(rng.integers(60, 900))
        a = rnd(n)
        b = list(a)
        for i in range(len(b)):
            if rng.random() < float(rng.choice([0.0, 0.05, 0.3, 0.45, 0.8])):
                b[i] = "ACGT"[rng.integers(0, 4)]
        b = "".join(b)[:max(50, n - int(rng.integers(0, 30)))]
        t = "INS" if rng.random() < 0.8 else "DEL"
        pv = snfp.PopulationVariant("chr1", int(rng.integers(1000, 1200)), "x", a, t, n if t == "INS" else -n, 0, 0.1, 10, 2)
        sv = NS(pos=pv.pos + int(rng.integers(-1300, 1300)), svlen=(len(b) if t == "INS" else -len(b)), svtype=t, alt=b)
        pairs.append((pv, sv))
    got = snfp.match_batch(pairs, cfg, _lib=E.lib())
    exp = []
    for pv, sv in pairs:
        dist = abs(pv.pos - sv.pos) + abs(abs(pv.svlen) - abs(sv.svlen))
        minlen = float(min(abs(pv.svlen), abs(sv.svlen)))
        if dist > cfg.combine_match * math.sqrt(minlen) or dist > cfg.combine_match_max:
            exp.append(None); continue
        if pv.svtype == "INS" and cfg.combine_pctseq:
            d = oracle_mod.edit_distance(pv.alt.encode(), sv.alt.encode())
            if (pv.svlen - d) / pv.svlen <= cfg.combine_pctseq:
                exp.append(None); continue
        exp.append(dist)
    assert got == exp and sum(e is not None for e in exp) > 10 and sum(e is None for e in exp) > 10
    assert pairs[0][0].match(pairs[0][1], cfg, _lib=E.lib()) == exp[0]
    assert snfp.PopulationVariant._calculate_frequency({0: (0, 1, 9), 1: ('.', '.', 0), 2: (1, 1, 5)}) == (0.75, 2, 2)
