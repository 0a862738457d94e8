"""sniffles_amd.fasta.FastaFile: pysam's `fetch(contig[, start, end])` contract over plain / indexed / gzip FASTA (the reader behind
`config.reference` for `_mask_N_coverage`, leadprov.py:424-441, and the VCF writer's REF / ALT resolution)."""
import gzip

import numpy as np
import pytest

from sniffles_amd import fasta, soa


def _write(tmp_path, width, gz=False, fai=False):
    rng = np.random.default_rng(3)
    seqs = {"chrA": "".join(rng.choice(list("ACGTN"), 1234)), "chrB desc": "".join(rng.choice(list("acgtN"), 61)), "chrC": "ACGT" * 30}
    text, index, pos = "", [], 0
    for k, v in seqs.items():
        head = f">{k}\n"
        body = "".join(v[i:i + width] + "\n" for i in range(0, len(v), width))
        index.append((k.split()[0], len(v), pos + len(head), width, width + 1))
        text += head + body
        pos += len(head) + len(body)
    p = tmp_path / ("ref.fa.gz" if gz else "ref.fa")
    if gz:
        with gzip.open(p, "wb") as f:
            f.write(text.encode())
    else:
        p.write_text(text)
    if fai:
        (tmp_path / "ref.fa.fai").write_text("".join("\t".join(map(str, r)) + "\n" for r in index))
    return str(p), {k.split()[0]: v for k, v in seqs.items()}


@pytest.mark.parametrize("width,gz,fai", [(60, False, False), (7, False, True), (1000000, False, False), (50, True, False)])
def test_fetch_equals_slicing(tmp_path, width, gz, fai):
    path, seqs = _write(tmp_path, width, gz, fai)
    with fasta.FastaFile(path) as f:
        assert f.references == list(seqs)
        for c, s in seqs.items():
            assert f.fetch(c) == s and f.get_reference_length(c) == len(s)
            for a, b in [(0, 1), (5, 70), (59, 61), (60, 120), (len(s) - 3, len(s) + 50), (len(s), len(s) + 5), (17, 17)]:
                assert f.fetch(c, a, b) == s[a:b], (c, a, b)
        with pytest.raises(KeyError):
            f.fetch("chrZ")
        with pytest.raises(ValueError):
            f.fetch("chrA", 10, 5)


def test_paint_nmask_is_the_dense_paint():
    """soa.paint_nmask == the reference's dense statements (leadprov.py:431-441) on overlapping, unsorted regions."""
    rng = np.random.default_rng(5)
    L = 5000
    for trial in range(40):
        seq = rng.choice(np.frombuffer(b"ACGN", np.uint8), L, p=[0.3, 0.3, 0.3, 0.1])
        for _ in range(6):
            a = int(rng.integers(0, L - 50)); seq[a:a + int(rng.integers(1, 400))] = ord("N")
        s = seq.tobytes().decode()
        # every region gets its own version of the sequence (so that overwriting is visible)
        versions, regions = [], []
        for k in range(int(rng.integers(1, 6))):
            a = int(rng.integers(0, L - 1)); b = int(rng.integers(a, L + 1))
            v = np.array(seq)
            v[rng.integers(0, L, 300)] = ord("N") if k % 2 else ord("A")
            versions.append(v.tobytes().decode()); regions.append((a, b))
        calls = iter(versions)

        def fetch(contig, start=None, end=None, _it=calls):
            return next(_it)[start:end]
        mask = np.zeros(L, np.uint8)
        for (a, b), v in zip(regions, versions):
            mask[a:b] = np.frombuffer(v[a:b].encode(), np.uint8)
        exp = soa.nmask_intervals(mask)
        got = soa.paint_nmask(fetch, "c", regions, L)
        assert np.array_equal(got[0], exp[0]) and np.array_equal(got[1], exp[1])
        assert np.all(got[0][1:] > got[1][:-1])           # sorted, disjoint, not touching
    whole = soa.paint_nmask(lambda c, a=None, b=None: s, "c", None, L)
    assert np.array_equal(whole[0], soa.nmask_intervals(s)[0])
    with pytest.raises(IndexError):
        soa.paint_nmask(lambda c, a=None, b=None: s + "N", "c", None, L)
    with pytest.raises(ValueError):
        soa.paint_nmask(lambda c, a=None, b=None: s[a:b] + "NN", "c", [(0, 100)], L)


def test_gzip_with_fai_uses_the_index_and_scan_equals_fai(tmp_path):
    """A `.fai` next to a gzip FASTA is used (its offsets index the decompressed text); without one the vectorised scan builds the
    same table - also for "\r\n" line ends and a last line without a newline."""
    p, seqs = _write(tmp_path, 37, gz=True, fai=True)
    f = fasta.FastaFile(p)
    (tmp_path / "ref.fa.gz.fai").write_text((tmp_path / "ref.fa.fai").read_text())
    g = fasta.FastaFile(p)                      # with the index
    (tmp_path / "ref.fa.gz.fai").unlink()
    assert f._index == g._index                 # scan == .fai
    for name, seq in seqs.items():
        assert g.fetch(name) == seq and f.fetch(name, 5, 50) == seq[5:50]
    q = tmp_path / "crlf.fa"
    q.write_bytes(b">a x\r\nACGT\r\nAC\r\n>b\r\nGG")
    h = fasta.FastaFile(str(q))
    assert h.fetch("a") == "ACGTAC" and h.fetch("b") == "GG" and h._index["a"] == (6, 6, 4, 6)
