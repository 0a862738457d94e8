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
    E.lib()                                              # the host tier becomes the library of this test
    sizes = [1, 2, 63, 64, 65, 127, 128, 129, 300, 511, 513, 700] + list(np.random.default_rng(0).integers(1, 400, 60))
    pairs = make_pairs(1, sizes, alphabet=b"ACGTNacgt<>")
    got = lib.edit_distance_batch(pairs)
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
    E.lib()                                              # the host tier becomes the library of this test
    sizes = [1, 2, 63, 64, 65, 127, 128, 129, 300, 511, 513, 700, 1500] + list(np.random.default_rng(7).integers(1, 400, 40))
    pairs = make_pairs(8, sizes, alphabet=b"ACGTN")
    check_banded(pairs, [oracle_mod.edit_distance(a, b) for a, b in pairs])


def acgt_pairs(seed, sizes):
    """Pairs whose shorter string is over {A, C, G, T} (what the ACGT column step of the banded wave form takes: Peq selects, two-bit
    carries, eight columns per step): a mutated copy over the same letters, one with a few other bytes in it ('N', lower case: the text
    may hold anything), an unrelated string, the string itself, and every residue of the text's length modulo eight."""
    rng = np.random.default_rng(seed)
    al = list(b"ACGT")
    out = []
    for n in sizes:
        a = bytes(rng.choice(al, n).astype(np.uint8))
        b = bytearray(a)
        for _ in range(int(rng.integers(1, max(2, n // 12)))):
            p = int(rng.integers(0, len(b)))
            op = rng.integers(0, 3)
            if op == 0:
                b[p] = int(rng.choice(al))
            elif op == 1:
                b.insert(p, int(rng.choice(al)))
            elif len(b) > 1:
                del b[p]
        b = bytes(b)
        out.append((a, b))
        out.append((a, a))
        odd = bytearray(b + bytes(rng.choice(al, int(rng.integers(8, 40))).astype(np.uint8)))      # the longer one: the text
        for p in rng.integers(0, len(odd), 5):
            odd[int(p)] = int(rng.choice(list(b"Nacgt-")))
        out.append((a, bytes(odd)))
        out.append((bytes(odd), a))
        out.append((a, bytes(rng.choice(al, int(n + rng.integers(0, 60))).astype(np.uint8))))
        for r in range(8):                                                                         # the text's last group: 1 .. 8 columns
            out.append((a, b + bytes(rng.choice(al, (r - len(b)) % 8 + 8).astype(np.uint8))))
    return out


def test_edit_distance_acgt_wave_forms_emulated(oracle_mod):
    import emu.emu as E
    E.lib()
    pairs = acgt_pairs(11, [513, 519, 640, 1100, 2050])
    check_banded(pairs, [oracle_mod.edit_distance(a, b) for a, b in pairs])


@pytest.mark.gpu
def test_edit_distance_acgt_wave_forms_gpu(oracle_mod):
    pairs = acgt_pairs(12, [513, 519, 640, 1100, 2050, 4096, 4097, 5003, 6100, 9000] + list(np.random.default_rng(9).integers(513, 7000, 30)))
    check_banded(pairs, [oracle_mod.edit_distance(a, b) for a, b in pairs])


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


def _population_pairs(seed=3, n_pairs=160):
    from types import SimpleNamespace as NS
    rng = np.random.default_rng(seed)

    def rnd(n):
        return "".join("ACGT"[i] for i in rng.integers(0, 4, n))
    out = []
    for _ in range(n_pairs):
        n = int(rng.integers(60, 900)) if rng.random() < 0.9 else int(rng.integers(900, 7000))
        a = rnd(n)
        b = list(a)
        for i in range(len(b)):
            if rng.random() < float(rng.choice([0.0, 0.05, 0.3, 0.45, 0.8])):
                b[i] = "ACGT"[rng.integers(0, 4)]
        b = "".join(b)[:max(50, n - int(rng.integers(0, 30)))]
        t = "INS" if rng.random() < 0.8 else "DEL"
        fields = dict(contig="chr1", pos=int(rng.integers(1000, 1200)), id="x", alt=a, svtype=t, svlen=n if t == "INS" else -n, end=0,
                      af=0.1, genotyped_sample_count=10, variant_sample_count=2)
        call = NS(pos=fields["pos"] + int(rng.integers(-1300, 1300)), svlen=(len(b) if t == "INS" else -len(b)), svtype=t, alt=b)
        out.append((fields, call))
    return out


def _check_population_match(L, extra_args=()):
    """`snfp.match_batch` against the UNMODIFIED reference's own `PopulationVariant.match` (`snfp.py:91-107`), its `align`
    (edlib, absent here) patched to the exact global unit-cost DP - edlib's default mode="NW", task="distance"; parity against
    edlib itself stays unpinned.  The option values (`--combine-match`, `--combine-match-max`,
    `--combine-pctseq`) come from the reference's own parser."""
    import ref_harness as rh
    import oracle
    from sniffles_amd import snfp
    rh.load_reference()
    from sniffles import snfp as ref_snfp
    cfg = rh.make_config(tuple(extra_args))            # (SnifflesConfig.__init__ makes it SnifflesConfig.GLOBAL, config.py:619)
    keep = ref_snfp.align
    ref_snfp.align = lambda a, b: {"editDistance": oracle.edit_distance(a.encode("latin-1"), b.encode("latin-1"))}
    try:
        pairs = _population_pairs()
        exp = [ref_snfp.PopulationVariant(**f).match(call) for f, call in pairs]
    finally:
        ref_snfp.align = keep
    mine = [(snfp.PopulationVariant(**f), call) for f, call in pairs]
    got = snfp.match_batch(mine, cfg)
    assert got == exp and sum(e is not None for e in exp) > 10 and sum(e is None for e in exp) > 10
    assert mine[0][0].match(mine[0][1], cfg) == exp[0]
    assert snfp.PopulationVariant._calculate_frequency({0: (0, 1, 9), 1: ('.', '.', 0), 2: (1, 1, 5)}) == \
        tuple(ref_snfp.PopulationVariant._calculate_frequency({0: (0, 1, 9), 1: ('.', '.', 0), 2: (1, 1, 5)})) == (0.75, 2, 2)


needs_ref = pytest.mark.skipif(not __import__("make_ref").ref_root(), reason="needs the reference (its checkout, or the staged build oracle/_ref that make_ref.py compiles)")


@needs_ref
@pytest.mark.parametrize("extra", [(), ("--combine-pctseq", "0.9", "--combine-match", "100"), ("--combine-pctseq", "0")])
def test_population_variant_match_equals_the_reference_emu(oracle_mod, extra):
    import emu.emu as E
    _check_population_match(E.lib(), extra)


@needs_ref
@pytest.mark.gpu
def test_population_variant_match_equals_the_reference_gpu(oracle_mod):
    _check_population_match(None)
    _check_population_match(None, ("--combine-pctseq", "0.9", "--combine-match", "100"))


def test_oracle_myers_stand_in_equals_the_exact_dp():
    """oracle/snf_oracle.c::snf_oracle_edit_distance_myers - the bit-parallel algorithm edlib implements, the edlib stand-in of the config-4
    reference baseline (bench.py --config 4) - against the exact DP: lengths around the 64-row word boundaries, near and unrelated strings."""
    import random
    import oracle as oc
    oc.build()
    rng = random.Random(77)
    for _ in range(1500):
        la = rng.choice([0, 1, 2, 5, 17, 63, 64, 65, 100, 127, 128, 129, 200, 511])
        a = bytes(rng.choice(b"ACGT") for _ in range(la))
        if rng.random() < 0.5 and la:
            b = bytearray(a)
            for _ in range(rng.randint(0, max(1, la // 8))):
                op, p = rng.random(), rng.randrange(len(b) + 1)
                if op < 0.33 and b:
                    b[min(p, len(b) - 1)] = rng.choice(b"ACGT")
                elif op < 0.66:
                    b.insert(p, rng.choice(b"ACGT"))
                elif b:
                    del b[min(p, len(b) - 1)]
            b = bytes(b)
        else:
            b = bytes(rng.choice(b"ACGTN") for _ in range(rng.choice([0, 1, 3, 64, 65, 127, 128, 300])))
        assert oc.edit_distance_myers(a, b) == oc.edit_distance(a, b), (a, b)
