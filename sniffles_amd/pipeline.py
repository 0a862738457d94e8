"""One sample from a BAM to VCF / SNF on one GPU: the `call_sample` flow of the reference's main program with
`--threads`-independent results (`src/sniffles/sniffles:286-360, 487-560`, `CallTask.execute` `parallel.py:255-297`),
wired from the pieces of this package - nothing here computes:

    BGZF inflate + header (host, sniffles_amd.bam)
    -> per contig task: signature extraction (GPU, sniffles_amd.extract) -> clustering / calling / QC / genotyping /
       consensus (GPU, sniffles_amd.parallel.CallTask) -> calls sorted by position
    -> VCF records in task order (sniffles_amd.vcf), SNF part files concatenated behind one header (sniffles_amd.snf)

Task layout as in the reference with its default `task_count_multiplier = 0`: one task per processed contig, ids in
header order, region [0, contig_length - 1), read ids offset by `task.id * 10 ** ceil(ln(total_mapped) + 1)`.  This is
the configs[0]-shaped plumbing path (BASELINE.json) and the end-to-end parity harness; worker processes, the CLI and
progress reporting stay with the reference.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

from . import bam, extract, parallel, snf, vcf


def should_process_contig(contig: str, length: int, config) -> bool:
    """`util.should_process_contig` (util.py:147-162): requested contigs / regions, otherwise contigs of 1 Mb and more."""
    wanted = getattr(config, "contig", None)
    by_region = getattr(config, "regions_by_contig", None) or {}
    if wanted and contig not in wanted:
        return False
    if by_region and contig not in by_region:
        return False
    if not getattr(config, "all_contigs", False) and length < 1_000_000:
        return bool((wanted and contig in wanted) or (contig in by_region))
    return True


@dataclass
class SampleResult:
    contig_lengths: list
    read_count: int = 0
    vcf_records: int = 0
    snf_candidates: int = 0
    calls: dict = field(default_factory=dict)     # task id -> sorted calls (as written)


class _Extracted:
    """Lead-provider stand-in for a task whose input came from the extraction kernels: the TaskInput is ready."""

    def __init__(self, ti):
        self.ti = ti
        self.contig_len, self.end = ti.contig_len, None
        self.device_batch = None

    def to_task_input(self, task_id, sv_id_start, tandem_repeats, qc_nm_threshold):
        return self.ti


def regions_of(config, contig: str) -> list:
    """`config.regions_by_contig.get(contig)` as [(start, end), ...] in list order; entries may be the reference's `Region`
    objects (contig, start, end attributes) or (contig, start, end) / (start, end) tuples, one or a list of them."""
    r = (getattr(config, "regions_by_contig", None) or {}).get(contig)
    if not r:
        return []
    if not isinstance(r, list):
        r = [r]
    out = []
    for x in r:
        if hasattr(x, "start"):
            out.append((int(x.start), int(x.end)))
        else:
            x = tuple(x)
            out.append((int(x[-2]), int(x[-1])))
    return out


def open_reference(config, reference=None):
    """The FASTA handle `_mask_N_coverage` reads (leadprov.py:424-429: `pysam.FastaFile(config.reference)`; a failure to open is
    logged and the sample goes on unmasked).  `reference`: an object with pysam's `fetch(contig[, start, end])`, else
    `config.reference` is opened with this package's plain reader (sniffles_amd.fasta)."""
    if reference is not None:
        return reference
    path = getattr(config, "reference", None)
    if not path or not isinstance(path, str):
        return None
    import logging
    from . import fasta
    try:
        return fasta.FastaFile(path)
    except Exception as e:  # noqa: BLE001 - as the reference: warn, no mask
        logging.warning(f"Unable to mask N regions in coverage vector, reference could not be opened: {e}")
        return None


def mask_N_coverage(ti, fasta_handle, contig: str, regions) -> None:
    """`LeadProvider._mask_N_coverage(regions)` for a task input that came from the extraction kernels: the task's coverage
    reads as 0 where the reference base is 'N' (`build_leadtab` calls it unconditionally once the regions are read,
    leadprov.py:470).  `regions`: the task's [(start, end)] in list order - `build_leadtab` always passes a list, the whole task
    being `[Region(contig, task.start, task.end)]` (parallel.py:101).  Failures leave the task unmasked with the reference's warning."""
    if fasta_handle is None:
        return
    import logging
    from . import soa
    try:
        ti.nmask_start, ti.nmask_end = soa.paint_nmask(fasta_handle.fetch, contig, regions, int(ti.contig_len))
    except Exception as e:  # noqa: BLE001
        ti.nmask_start = ti.nmask_end = None
        logging.warning(f"Unable to mask N regions in coverage vector, reference could not be fetched: {e}")


class _RegionExtractor:
    """Stands in for the extractor of a device-resident task when the task came from a region list (host concatenation)."""

    def close(self):
        pass


def _extract_regions(recs, contig, regions, config, read_id_offset, task_id, tandem_repeats, device):
    """`build_leadtab(regions, bam)` (leadprov.py:445-472): one extraction per region, in list order, into one task; the
    running read id carries on from region to region."""
    from . import soa
    parts, read_count, rid = [], 0, int(read_id_offset)
    info = None
    for start, end in regions:
        ti, info = extract.extract_region(recs, contig, start, end, config, read_id_offset=rid % 2 ** 32, task_id=task_id, sv_id_start=0,
                                          tandem_repeats=tandem_repeats, device=device)
        parts.append(ti)
        read_count += info.read_count
        rid = info.read_id
    info.read_count = read_count
    return soa.concat_tasks(parts), info, _RegionExtractor()


def call_sample(records: bam.BamRecords, config, vcf_handle=None, snf_path=None, tandem_repeats=None, device: int = 0, objects: bool = True, reference=None) -> SampleResult:
    """`records`: `bam.read_bam(path)`.  `tandem_repeats`: {contig: [(start, end), ...]} (already padded, util.py:121-144).
    Writes the VCF to `vcf_handle` and / or the SNF to `snf_path` (CallTask.execute switches QC filtering off for the
    candidates when an SNF is requested, parallel.py:258-263).
    `objects=False`: VCF only, formatted straight from the record table (vcf.VCF.write_records) - the same text, no `SVCall`
    objects (`SampleResult.calls` stays empty); falls back to the object path when a reference FASTA is attached.
    `reference` / `config.reference`: the reference FASTA (see `open_reference`) - with it the coverage of every task is masked
    where the reference base is 'N' (`_mask_N_coverage`, leadprov.py:420-443, 470) and the writer resolves REF / ALT."""
    import struct
    flags = [struct.unpack_from("<H", records.blob, int(o) + 18)[0] for o in records.rec_off[:-1]]
    total_mapped = sum(1 for f, r in zip(flags, records.ref_id.tolist()) if r >= 0 and not f & 0x4)
    config.task_read_id_offset_mult = 10 ** 9 if total_mapped == 0 else 10 ** math.ceil(math.log(total_mapped) + 1)
    config.snf = snf_path
    contig_lengths = [(c, int(n)) for c, n in zip(records.ref_names, records.ref_lens) if should_process_contig(c, int(n), config)]
    config.contig_lengths = contig_lengths
    out = SampleResult(contig_lengths=contig_lengths)
    writer = None
    fasta_handle = open_reference(config, reference) if (reference is not None or getattr(config, "reference", None)) else None
    if vcf_handle is not None:
        writer = vcf.VCF(config, vcf_handle)
        if fasta_handle is not None and writer.reference_handle is None:
            writer.reference_handle = fasta_handle          # (vcf.py:108-120 open_reference: the same file)
        writer.write_header(contig_lengths)
    snf_out = snf.SNFile(config, open(snf_path, "wb")) if snf_path else None
    qc = not (snf_path is not None or config.no_qc)
    for task_id, (contig, length) in enumerate(contig_lengths):
        tr = (tandem_repeats or {}).get(contig)
        task = parallel.CallTask(id=task_id, sv_id=0, contig=contig, start=0, end=length - 1, config=config,
                                 tandem_repeats=tr, device=device)
        # the signatures never leave HBM between the extraction and the clustering batch (snf_batch_add_task_device)
        regions = regions_of(config, contig)
        if regions:      # --regions: the task's leads and coverage come from these intervals only (sniffles:330-341)
            ti, info, extractor = _extract_regions(bam.contig_records(records, contig), contig, regions, config,
                                                   (task_id * config.task_read_id_offset_mult) % 2 ** 32, task_id, tr, device)
        else:
            ti, info, extractor = extract.extract_region_device(bam.contig_records(records, contig), contig, task.start, task.end, config,
                                                                read_id_offset=(task_id * config.task_read_id_offset_mult) % 2 ** 32,
                                                                task_id=task_id, sv_id_start=0, tandem_repeats=tr, device=device)
        config.qc_nm_threshold = config.average_regional_nm = ti.qc_nm_threshold      # iter_region's side channel
        mask_N_coverage(ti, fasta_handle, contig, regions or [(task.start, task.end)])
        task.lead_provider = _Extracted(ti)
        if not objects and snf_out is None and writer is not None and writer.can_write_records():
            import numpy as np
            try:
                # CallTask.execute's QC filter and position sort happen on the device (SNF_OUT_EXECUTE): only the records the
                # VCF will hold cross PCIe, already in output order
                res, ti_used = task.call_records(config, execute=True)
                out.vcf_records += writer.write_records(res, ti_used, np.arange(len(res.calls)))
                out.read_count += info.read_count
            finally:
                task.close()
                extractor.close()      # (backs the lazy host copies of the task input: goes after the task)
            continue
        try:
            cands = task.call_candidates(qc, config)
            calls = task.finalize_candidates(cands, not qc, config)
            if not config.no_qc:
                calls = [c for c in calls if c.qc]
            if getattr(config, "sort", True):
                calls = sorted(calls, key=lambda c: c.pos)
            out.read_count += info.read_count
            if snf_out is not None:
                snf_out.add_result(task.write_snf_part(cands, f"{snf_path}.tmp_{task_id}.snf"))
        finally:
            task.close()
            extractor.close()
        if writer is not None:
            out.vcf_records += sum(writer.write_call(c) for c in calls)
        out.calls[task_id] = calls
    if snf_out is not None:
        out.snf_candidates = snf_out.write_results(config, [c for c, _ in contig_lengths])
        snf_out.close()
    return out


from . import candstore  # noqa: E402


def combine(snf_paths, config, vcf_handle=None, sample_ids=None, device: int = 0, objects: bool = True) -> list:
    """Multi-sample calling from per-sample `.snf` files: the `combine` flow of the reference's main program
    (`sniffles:371-490`) for one process - headers (sample ids, contig lengths, format checks), one `CombineTask` per
    contig over `snf.SNFile` readers (group assignment on the GPU), calls of a task sorted by position
    (`CombineResult`), VCF records in task order.  Returns the combined calls.  `objects=False`: VCF only, the records formatted
    straight from the group table when the writer can (`VCF.can_write_merged`) - the same text; returns []."""
    import os
    config.mode = "combine"
    config.snf_input_info, readers = [], {}
    contig_lengths = None
    for internal_id, path in enumerate(snf_paths):
        f = snf.SNFile(config, open(path, "rb"), filename=path)
        f.read_header()
        hc = f.header["config"]
        if config.snf_block_size != hc["snf_block_size"]:
            raise ValueError(f"SNF block size differs for {path}")
        if config.snf_format_version != hc["snf_format_version"]:
            raise ValueError(f"SNF format version for {path} is not supported")
        contig_lengths = hc["contig_lengths"]                 # the last header wins, like in the reference
        sid = (sample_ids[internal_id] if sample_ids else None) or hc.get("sample_id") or os.path.splitext(os.path.basename(path))[0]
        config.snf_input_info.append({"internal_id": internal_id, "sample_id": sid, "filename": path})
        readers[internal_id] = f
    config.sample_ids_vcf = [(i["internal_id"], i["sample_id"]) for i in config.snf_input_info]
    contig_lengths = [(c, int(n)) for c, n in (contig_lengths or [])]
    wanted = getattr(config, "contig", None) or getattr(config, "regions_by_contig", None)
    if wanted:
        contig_lengths = [(c, n) for c, n in contig_lengths if c in wanted]
    writer = None
    if vcf_handle is not None:
        writer = vcf.VCF(config, vcf_handle)
        writer.write_header(contig_lengths)
    out = []
    tasks = [parallel.CombineTask(id=task_id, sv_id=0, contig=contig, start=0, end=length - 1, config=config, device=device,
                                  regions=(getattr(config, "regions_by_contig", None) or {}).get(contig))
             for task_id, (contig, length) in enumerate(contig_lengths)]
    if not objects and writer is not None and writer.can_write_merged():
        # VCF only: the merged records are formatted straight from the group table (vcf.VCF.write_merged) - the same text, no SVCall objects
        for part in parallel.CombineTask.execute_many(tasks, readers, text_writer=writer):
            writer.write_merged(part, sort=getattr(config, "sort", True))
        for f in readers.values():
            candstore.clear_columns(f)      # (the merge is written: the readers' column tables go with it)
            f.close()
        return []
    # all contigs share one group-assignment launch (a contig alone leaves most of the device idle)
    for task, calls in zip(tasks, parallel.CombineTask.execute_many(tasks, readers)):
        if getattr(config, "sort", True):
            calls = sorted(calls, key=lambda c: c.pos)
        if writer is not None:
            for c in calls:
                writer.write_call(c)
        out.extend(calls)
    for f in readers.values():
        candstore.clear_columns(f)
        f.close()
    return out


def genotype_vcf(records: bam.BamRecords, config, vcf_in_handle, vcf_out_handle, tandem_repeats=None, device: int = 0, reference=None) -> int:
    """Force calling (`--genotype-vcf`, sniffles:190-213, 487-560 and `GenotypeTask.execute`): the SVs of the input VCF are
    matched against this sample's candidates contig by contig and written back with the sample's genotype (contig by
    contig, input order within a contig).  Returns the number of records written."""
    import struct
    config.mode = "genotype_vcf"
    reader = vcf.VCF(config, vcf_in_handle)
    by_contig = {}
    for target in reader.read_svs_iter():
        by_contig.setdefault(target.contig, []).append(target)
    flags = [struct.unpack_from("<H", records.blob, int(o) + 18)[0] for o in records.rec_off[:-1]]
    total_mapped = sum(1 for f, r in zip(flags, records.ref_id.tolist()) if r >= 0 and not f & 0x4)
    config.task_read_id_offset_mult = 10 ** 9 if total_mapped == 0 else 10 ** math.ceil(math.log(total_mapped) + 1)
    contig_lengths = [(c, int(n)) for c, n in zip(records.ref_names, records.ref_lens) if should_process_contig(c, int(n), config)]
    config.contig_lengths = contig_lengths
    writer = vcf.VCF(config, vcf_out_handle)
    writer.rewrite_header_genotype(reader.header_str)
    fasta_handle = open_reference(config, reference) if (reference is not None or getattr(config, "reference", None)) else None
    n = 0
    for task_id, (contig, length) in enumerate(contig_lengths):
        tr = (tandem_repeats or {}).get(contig)
        targets = [t for t in by_contig.get(contig, []) if 0 <= t.pos < length - 1]
        task = parallel.GenotypeTask(id=task_id, sv_id=0, contig=contig, start=0, end=length - 1, config=config, tandem_repeats=tr,
                                     genotype_svs=targets, device=device)
        regions = regions_of(config, contig) or [(task.start, task.end)]       # --regions: leads and coverage from these intervals only
        ti, _, _ = _extract_regions(bam.contig_records(records, contig), contig, regions, config,
                                    (task_id * config.task_read_id_offset_mult) % 2 ** 32, task_id, tr, device)
        config.qc_nm_threshold = config.average_regional_nm = ti.qc_nm_threshold
        mask_N_coverage(ti, fasta_handle, contig, regions)
        task.lead_provider = _Extracted(ti)
        res = task.execute()
        task.close()
        for target in res or []:      # task by task, input order inside a task (GenotypeResult.emit, result.py:118-131);
            writer.rewrite_genotype(target)      # targets on contigs that are not processed are dropped, like the reference
            n += 1
    return n
