"""Signature extraction on the GPU: the drop-in for `LeadProvider.build_leadtab` (reference `leadprov.py:445-472`).

`extract_region(recs, contig, start, end, config, ...)` hands the inflated BAM alignment records of one contig
(`sniffles_amd.bam.BamRecords`) to `snf_extract_*` (include/sniffles_amd.h, csrc/snf_extract.hip) and returns the
task input the clustering path consumes (`sniffles_amd.soa.TaskInput`): the leads in `record_lead` order, the
sequence pool, the per-read (start, end, HP) table that replaces the dense coverage vector and `record_hap_ref`,
and `config.qc_nm_threshold`.  The host part here is string bookkeeping only: read names and contig names become
order-preserving ranks, PS values come back as a rank table.  No CPU fallback: without the HIP library this raises.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import abi, bam
from .soa import LEAD_FIELDS, TaskInput


@dataclass
class ExtractInfo:
    read_id: int            # LeadProvider.read_id after the region
    read_count: int
    ms_count: float         # kernel time of the two passes (HIP events)
    ms_emit: float
    algo_bytes: int


class Extractor:
    """One extraction handle (device allocations are reused per upload)."""

    def __init__(self, config=None, device: int = 0):
        from . import lib as L
        self.lib = L.load()
        self._err = L.SnifflesAmdError
        self._h = C.c_void_p()
        cs = abi.extract_config_struct(config if config is not None else object())
        self._check(self.lib.snf_extract_create(C.byref(cs), device, C.byref(self._h)))
        self._keep = None

    def _check(self, rc):
        if rc != 0:
            raise self._err(self.lib.snf_extract_last_error().decode("utf-8", "replace"))

    def close(self):
        if self._h:
            self.lib.snf_extract_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, recs: bam.BamRecords, contig: str, start: int, end: int, read_id_offset: int = 0, qranks=None):
        """Host -> HBM.  `qranks`: (rank per record, sorted names) if the caller interned the read names already."""
        rank, names = qranks if qranks is not None else bam.qname_ranks(recs)
        hashes, hrank, rank_of_refid, contig_names = bam.contig_tables(recs.ref_names)
        rid = recs.ref_names.index(contig)
        blob = np.ascontiguousarray(recs.blob, np.uint8)
        off = np.ascontiguousarray(recs.rec_off, np.int64)
        rank = np.ascontiguousarray(rank, np.uint32)
        inp = abi.snf_extract_input_t(
            records=blob.ctypes.data_as(abi.u8p), records_len=int(blob.shape[0]),
            rec_off=off.ctypes.data_as(C.POINTER(C.c_int64)), n_records=recs.n,
            qname_rank=rank.ctypes.data_as(C.POINTER(C.c_uint32)), region_ref_id=rid,
            region_rank=int(rank_of_refid[rid]), region_start=int(start), region_end=int(end),
            read_id_offset=int(read_id_offset), n_contigs=len(recs.ref_names),
            contig_hash=hashes.ctypes.data_as(C.POINTER(C.c_uint64)), contig_rank=hrank.ctypes.data_as(C.POINTER(C.c_int32)))
        self._keep = (blob, off, rank, hashes, hrank)
        self._check(self.lib.snf_extract_upload(self._h, C.byref(inp)))
        self._ctx = dict(contig=contig, contig_len=int(recs.ref_lens[rid]), qnames=names, contig_names=contig_names)

    def run(self):
        self._check(self.lib.snf_extract_run(self._h))

    def host_columns(self) -> dict:
        """The result columns copied to the host (the first call after a run makes the copies)."""
        r = abi.snf_extract_result_t()
        self._check(self.lib.snf_extract_result(self._h, C.byref(r)))
        t = r.task
        n, npool, nr = int(t.n_leads), int(t.seq_pool_len), int(t.n_reads)
        leads = {name: (np.ctypeslib.as_array(getattr(t, name), shape=(n,)).astype(dt, copy=True) if n else np.zeros(0, dt))
                 for name, dt in LEAD_FIELDS}
        return dict(leads=leads,
                    seq_pool=np.ctypeslib.as_array(t.seq_pool, shape=(npool,)).copy() if npool else np.zeros(0, np.uint8),
                    read_start=np.ctypeslib.as_array(t.read_start, shape=(nr,)).copy() if nr else np.zeros(0, np.int32),
                    read_end=np.ctypeslib.as_array(t.read_end, shape=(nr,)).copy() if nr else np.zeros(0, np.int32),
                    read_hp=np.ctypeslib.as_array(t.read_hp, shape=(nr,)).copy() if nr else np.zeros(0, np.uint8))

    def result_device(self, task_id: int = 0, sv_id_start: int = 0, tandem_repeats=None):
        """The result as a task that STAYS in HBM (soa.DeviceTaskInput): `lib.Batch` takes its columns device-to-device.
        This extractor must stay open until the batch has been created.  Returns (DeviceTaskInput, ExtractInfo)."""
        from .soa import DeviceTaskInput
        r = abi.snf_extract_result_t()
        self._check(self.lib.snf_extract_result_meta(self._h, C.byref(r)))
        t = r.task
        ps_vals = np.ctypeslib.as_array(r.ps_value, shape=(int(r.n_ps),)).tolist()
        ps_names = [str(x) for x in ps_vals]
        ps_names[int(t.ps_null_rank)] = "NULL"
        c = self._ctx
        ti = DeviceTaskInput(self, int(t.n_leads), int(t.n_reads), task_id=task_id, contig=c["contig"], contig_len=c["contig_len"],
                             sv_id_start=sv_id_start,
                             tr_start=None if tandem_repeats is None else np.array([x[0] for x in tandem_repeats], np.int32),
                             tr_end=None if tandem_repeats is None else np.array([x[1] for x in tandem_repeats], np.int32),
                             qc_nm_threshold=float(t.qc_nm_threshold), qnames=c["qnames"], ps_names=ps_names,
                             contig_names=c["contig_names"])
        info = ExtractInfo(read_id=int(r.read_id), read_count=int(r.read_count), ms_count=float(r.ms_count),
                           ms_emit=float(r.ms_emit), algo_bytes=int(r.algo_bytes))
        return ti, info

    def result(self, task_id: int = 0, sv_id_start: int = 0, tandem_repeats=None):
        r = abi.snf_extract_result_t()
        self._check(self.lib.snf_extract_result(self._h, C.byref(r)))
        t = r.task
        n = int(t.n_leads)
        leads = {}
        for name, dt in LEAD_FIELDS:
            p = getattr(t, name)
            leads[name] = np.ctypeslib.as_array(p, shape=(n,)).astype(dt, copy=True) if n else np.zeros(0, dt)
        npool, nr = int(t.seq_pool_len), int(t.n_reads)
        pool = np.ctypeslib.as_array(t.seq_pool, shape=(npool,)).copy() if npool else np.zeros(0, np.uint8)
        rs = np.ctypeslib.as_array(t.read_start, shape=(nr,)).copy() if nr else np.zeros(0, np.int32)
        re = np.ctypeslib.as_array(t.read_end, shape=(nr,)).copy() if nr else np.zeros(0, np.int32)
        rh = np.ctypeslib.as_array(t.read_hp, shape=(nr,)).copy() if nr else np.zeros(0, np.uint8)
        ps_vals = np.ctypeslib.as_array(r.ps_value, shape=(int(r.n_ps),)).tolist()
        ps_names = [str(x) for x in ps_vals]
        ps_names[int(t.ps_null_rank)] = "NULL"
        c = self._ctx
        ti = TaskInput(task_id=task_id, contig=c["contig"], contig_len=c["contig_len"], sv_id_start=sv_id_start, leads=leads,
                       seq_pool=pool, read_start=rs, read_end=re, read_hp=rh,
                       tr_start=None if tandem_repeats is None else np.array([x[0] for x in tandem_repeats], np.int32),
                       tr_end=None if tandem_repeats is None else np.array([x[1] for x in tandem_repeats], np.int32),
                       qc_nm_threshold=float(t.qc_nm_threshold), qnames=c["qnames"], ps_names=ps_names,
                       contig_names=c["contig_names"])
        ti.validate()
        info = ExtractInfo(read_id=int(r.read_id), read_count=int(r.read_count), ms_count=float(r.ms_count),
                           ms_emit=float(r.ms_emit), algo_bytes=int(r.algo_bytes))
        return ti, info


def extract_region(recs: bam.BamRecords, contig: str, start: int, end: int, config=None, read_id_offset: int = 0,
                   task_id: int = 0, sv_id_start: int = 0, tandem_repeats=None, device: int = 0):
    """`LeadProvider(config, read_id_offset, contig).build_leadtab([Region(contig, start, end)], bam)` on the GPU.
    Returns (TaskInput, ExtractInfo)."""
    x = Extractor(config, device)
    try:
        x.upload(recs, contig, start, end, read_id_offset)
        x.run()
        return x.result(task_id, sv_id_start, tandem_repeats)
    finally:
        x.close()


def extract_region_device(recs: bam.BamRecords, contig: str, start: int, end: int, config=None, read_id_offset: int = 0,
                          task_id: int = 0, sv_id_start: int = 0, tandem_repeats=None, device: int = 0):
    """`extract_region` whose result stays in HBM: returns (soa.DeviceTaskInput, ExtractInfo, Extractor).  The clustering
    batch takes the columns device-to-device; close the extractor once the task is through (its memory backs the task until
    the batch has been created, and the lazy host copies of the columns afterwards)."""
    x = Extractor(config, device)
    try:
        x.upload(recs, contig, start, end, read_id_offset)
        x.run()
        ti, info = x.result_device(task_id, sv_id_start, tandem_repeats)
        return ti, info, x
    except Exception:
        x.close()
        raise
