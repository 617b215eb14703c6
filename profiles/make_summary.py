#!/usr/bin/env python3
"""Condenses rocprofv3 output directories (gpurun_out/...) into the small, committed summaries under profiles/.
usage: make_summary.py TAG KERNEL_TRACE_DIR FETCH_PMC_DIR WRITE_PMC_DIR [SQ_PMC_DIR|-] [REQ_PMC_DIR|-] [WORKLOAD]"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from elba_amd.capi import numeric_source_fingerprint      # noqa: E402 — what the counter summaries are valid for (bench.py checks it)


def short(n):
    return re.sub(r"elba::\(anonymous namespace\)::", "", n).split("(")[0].replace("void ", "")


NUMERIC = ("k_spgemm_direct", "k_spgemm_rows")      # the numeric kernels (plan-free; round-1 descriptor kernel)


def main():
    tag, ktd, fd, wd = sys.argv[1:5]
    sqd = sys.argv[5] if len(sys.argv) > 5 and sys.argv[5] != "-" else None
    rqd = sys.argv[6] if len(sys.argv) > 6 and sys.argv[6] != "-" else None
    workload = sys.argv[7] if len(sys.argv) > 7 else "200k-long-reads"
    ks = max(glob.glob(os.path.join(ktd, "*", "*kernel_stats.csv")), key=os.path.getmtime)      # (gpurun_out/ keeps earlier runs of the same tag: the newest)
    shutil.copy(ks, os.path.join(HERE, "%s_kernel_stats.csv" % tag))
    rows = list(csv.DictReader(open(ks)))
    stats = {short(r["Name"]): dict(calls=int(r["Calls"]), avg_us=float(r["AverageNs"]) / 1e3, total_ms=float(r["TotalDurationNs"]) / 1e6) for r in rows}
    calls = max(v["calls"] for k, v in stats.items() if k.startswith("k_finalize_wave"))      # one finalize launch per step
    spg = {k: v for k, v in stats.items() if k.startswith(NUMERIC)}
    numeric_us_per_step = sum(v["total_ms"] for v in spg.values()) * 1e3 / calls

    def pmc(d):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        n = collections.Counter()
        f = max(glob.glob(os.path.join(d, "*", "*counter_collection.csv")), key=os.path.getmtime)
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[(k, r["Counter_Name"])] += 1
        return agg, n

    fa, fn = pmc(fd)
    wa, wn = pmc(wd)
    steps_f = fn[("k_finalize_wave", "FETCH_SIZE")]
    steps_w = wn[("k_finalize_wave", "WRITE_SIZE")]
    fetch_kb = sum(v["FETCH_SIZE"] for k, v in fa.items() if k.startswith(NUMERIC)) / steps_f
    write_kb = sum(v["WRITE_SIZE"] for k, v in wa.items() if k.startswith(NUMERIC)) / steps_w
    hit = sum(v.get("TCC_HIT_sum", 0) for k, v in wa.items() if k.startswith(NUMERIC)) / steps_w
    miss = sum(v.get("TCC_MISS_sum", 0) for k, v in wa.items() if k.startswith(NUMERIC)) / steps_w
    out = {
        "tag": tag, "workload": workload, "n_gpus": 1,
        "steps_in_trace": calls,
        "numeric_kernels_us_per_step": round(numeric_us_per_step, 2),
        "kernel_avg_us": {k: round(v["avg_us"], 2) for k, v in sorted(stats.items()) if k.startswith("k_")},
        "FETCH_SIZE_KB_per_step": round(fetch_kb, 1), "WRITE_SIZE_KB_per_step": round(write_kb, 1),
        "TCC_HIT_per_step": round(hit), "TCC_MISS_per_step": round(miss),
        "bytes_per_TCC_miss": round(fetch_kb * 1024 / max(1, miss), 1),
        # MI355X_MICROARCH.md §HBM: FETCH_SIZE = TCC_EA0_RDREQ x 64 B and under-reports 128-B streaming requests by 2x.  Calibrated on this
        # kernel's own widths (profiles/microbench/fetchcal.hip, r03_fetchcal.txt): a coalesced stream is fetched in 128-byte requests at 4, 8
        # and 16 bytes per lane alike, each tallied at 64 bytes.  The kernel's one coalesced stream is the row entries of A (8 bytes each, read
        # once per step): FETCH_SIZE misses half of it, 4 bytes per entry, which is added back; its gathers are 64-byte column rows, counted
        # in full.  The face value and the figure with EVERY read doubled are kept beside the corrected one.
        "hbm_bytes_per_step_face_value": int(fetch_kb * 1024 + write_kb * 1024),
        "hbm_bytes_per_step_upper_bound_if_128B_requests": int(2 * fetch_kb * 1024 + write_kb * 1024),
    }
    nnz_a = None
    for cand in (os.path.join(HERE, "%s_bench_line.json" % tag), os.path.join(os.path.dirname(HERE), "gpurun_out", "%s_bench.json" % tag)):
        if os.path.exists(cand):
            try:
                nnz_a = int(json.loads(open(cand).read().strip().splitlines()[-1])["config"]["nnz_A"])
                break
            except Exception:
                pass
    out["row_entry_stream_correction_bytes"] = 4 * nnz_a if nnz_a else None
    out["hbm_bytes_per_step_dominant_kernel"] = out["hbm_bytes_per_step_face_value"] + (4 * nnz_a if nnz_a else 0)
    out["hbm_bytes_note"] = ("FETCH_SIZE + 4 B per row entry of A (the coalesced row-entry stream is fetched in 128-byte requests tallied at 64: r03_fetchcal.txt) + WRITE_SIZE"
                             if nnz_a else "face value: nnz(A) of the workload not found, the row-entry stream is NOT corrected")
    if sqd:
        sa, sn = pmc(sqd)
        steps_s = sn[("k_finalize_wave", "SQ_WAVES")] or 1
        out["SQ_per_step"] = {c: round(sum(v.get(c, 0) for k, v in sa.items() if k.startswith(NUMERIC)) / steps_s)
                              for c in ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS")}
    if rqd:
        ra, rn = pmc(rqd)
        steps_r = rn[("k_finalize_wave", "TCC_REQ_sum")] or 1
        req = {c: round(sum(v.get(c, 0) for k, v in ra.items() if k.startswith(NUMERIC)) / steps_r) for c in ("TCC_REQ_sum", "TCP_TCC_READ_REQ_sum", "TCP_TCC_WRITE_REQ_sum", "TCP_TCC_ATOMIC_WITH_RET_REQ_sum")}
        out["L2_requests_per_step"] = req
        json.dump({"workload": workload, "n_gpus": 1, "numeric_source_sha16": numeric_source_fingerprint(), "source": "%s_summary.json (rocprofv3 --pmc TCC_REQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum, numeric kernels of one step)" % tag,
                   "tcc_requests_per_step": req["TCC_REQ_sum"], "read_requests_per_step": req["TCP_TCC_READ_REQ_sum"], "write_requests_per_step": req["TCP_TCC_WRITE_REQ_sum"],
                   "returning_atomics_per_step": req["TCP_TCC_ATOMIC_WITH_RET_REQ_sum"],
                   "ceiling_source": "profiles/microbench/gather64.hip (profiles/r02_gather64_microbench.txt): random 64-byte lines read by 4 lanes x 16 B, 350 MB and 12.9 GB arrays",
                   "ceiling_G_requests_per_s": [46.5, 52.5]}, open(os.path.join(HERE, "requests.json"), "w"), indent=1)
    json.dump(out, open(os.path.join(HERE, "%s_summary.json" % tag), "w"), indent=1)
    json.dump({"workload": out["workload"], "n_gpus": 1, "numeric_source_sha16": numeric_source_fingerprint(), "source": "%s_summary.json" % tag,
               "hbm_bytes_per_step_dominant_kernel": out["hbm_bytes_per_step_dominant_kernel"], "hbm_bytes_per_step_face_value": out["hbm_bytes_per_step_face_value"],
               "note": out["hbm_bytes_note"]}, open(os.path.join(HERE, "traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
