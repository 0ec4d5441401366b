#!/usr/bin/env python3
"""Per-kernel summary of three rocprofv3 --pmc passes over tools/prof_frame.py:

  rocprofv3 --kernel-trace --pmc FETCH_SIZE  --output-format csv -d gpurun_out/pmc_fetch -o p -- python tools/prof_frame.py M2 6
  rocprofv3 --kernel-trace --pmc WRITE_SIZE  --output-format csv -d gpurun_out/pmc_write -o p -- python tools/prof_frame.py M2 6
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_F32 \
            SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_sq -o p -- ...

HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE reports half of wide coalesced reads,
MI355X_MICROARCH.md section HBM; counter unit KB).  MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES /
(GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs).  Usage: python tools/pmc_summary.py <tag>  ->
profiles/<tag>_pmc_summary.csv and profiles/pmc_dominant_kernel.json (read by bench.py)."""
import collections
import csv
import json
import statistics as st
import sys


def load(path):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        d[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return d


def durs(path):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        d[r["Kernel_Name"].split("(")[0]].append(float(r["End_Timestamp"]) -
                                                 float(r["Start_Timestamp"]))
    return d


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    # "train": the passes of tools/profile_train.sh over the batch-8 training step (every kernel
    # of the step, library GEMMs and torch's elementwise kernels included)
    train = len(sys.argv) > 2 and sys.argv[2] == "train"
    base = f"gpurun_out/{tag}_train/" if train else "gpurun_out/"
    f = load(base + "pmc_fetch/p_counter_collection.csv")
    w = load(base + "pmc_write/p_counter_collection.csv")
    s = load(base + "pmc_sq/p_counter_collection.csv")
    du = durs(base + "pmc_sq/p_kernel_trace.csv")
    rows = []
    for k in sorted(s, key=lambda k: -sum(du.get(k, [0]))):
        if not train and ("at::" in k or "rocclr" in k or "Cijk" in k):
            continue
        fe = st.mean(f[k]["FETCH_SIZE"]) * 1024 if k in f else 0
        wr = st.mean(w[k]["WRITE_SIZE"]) * 1024 if k in w else 0
        d = st.mean(du[k])
        sq = {c: st.mean(v) for c, v in s[k].items()}
        cyc = sq["GRBM_GUI_ACTIVE"] / 8
        hbm = 2 * fe + wr
        wc = sq.get("SQ_WAVE_CYCLES", 0)
        rows.append(dict(
            kernel=k.replace("void ", ""), launches=len(du[k]), avg_us=round(d / 1e3, 2),
            fetch_size_bytes=int(fe), write_size_bytes=int(wr), hbm_bytes_corrected=int(hbm),
            hbm_GBps=round(hbm / d, 1), mfma_f32_insts=int(sq.get("SQ_INSTS_VALU_MFMA_F32", 0)),
            mfma_busy_cycles=int(sq.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)),
            mfma_util=round(sq.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cyc * 1024), 4) if cyc else 0,
            wait_any_frac=round(sq["SQ_WAIT_ANY"] / wc, 3) if wc else 0,
            wait_inst_frac=round(sq["SQ_WAIT_INST_ANY"] / wc, 3) if wc else 0))
    with open(f"profiles/{tag}{'_train' if train else ''}_pmc_summary.csv", "w") as fo:
        wri = csv.DictWriter(fo, fieldnames=list(rows[0].keys()))
        wri.writeheader()
        wri.writerows(rows)
    # the dominant kernel of bench.py: the 3x3x3 64->64 forward (quad-row form; before round 2's
    # end the tile-compacted one, one or two wave sets)
    if train:
        for r in rows[:40]:
            print(f"{r['kernel'][:60]:60s} x{r['launches']:4d} {r['avg_us']:8.2f} us  "
                  f"{r['hbm_GBps']:8.1f} GB/s  mfma_util {r['mfma_util']}")
        return
    tc = [r for r in rows if r["kernel"].startswith("cv_q4_kernel<64, 64")]
    name = "cv_q4_kernel<64,64,4>"
    if not tc:
        tc = [r for r in rows if r["kernel"].startswith("cv_tc_kernel<64, 64")]
        name = "cv_tc_kernel<64,64,32>"
    if tc:
        n = sum(r["launches"] for r in tc)
        avg = lambda key: sum(r[key] * r["launches"] for r in tc) / n
        json.dump({"kernel": name,
                   "hbm_bytes_per_launch": int(avg("hbm_bytes_corrected")),
                   "mfma_util": round(avg("mfma_util"), 4), "launches": n,
                   "avg_us": round(avg("avg_us"), 2),
                   "source": f"profiles/{tag}_pmc_summary.csv"},
                  open("profiles/pmc_dominant_kernel.json", "w"), indent=1)
    for r in rows:
        print(f"{r['kernel'][:40]:40s} {r['avg_us']:8.2f} us  {r['hbm_GBps']:8.1f} GB/s  "
              f"mfma_util {r['mfma_util']}")


if __name__ == "__main__":
    main()
