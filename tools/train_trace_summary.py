#!/usr/bin/env python3
"""Steady-state summary of the training-step profiles of tools/profile_train.sh.

On a fresh GPU box MIOpen searches its solvers during the first steps (naive reference kernels
of 50-200 ms, dozens of CK candidates), so whole-process rocprofv3 statistics are dominated by
that search.  This tool cuts the kernel trace into steps at the fused AdamW launch (the last
kernel group of a step) and keeps the LAST `steps` steps only:

  python tools/train_trace_summary.py <tag> [steps]     (default 10)

  gpurun_out/<tag>_train/kt/p_kernel_trace.csv        -> profiles/<tag>_train_kernel_stats.csv
  gpurun_out/<tag>_train/pmc_{fetch,write,sq}/*.csv   -> profiles/<tag>_train_pmc_summary.csv
                                                         (the last step of each counter pass)
HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (counter unit KB; gfx950 correction of
MI355X_MICROARCH.md, as tools/pmc_summary.py)."""
import collections
import csv
import statistics as st
import sys

STEP_END = "FusedAdamMathFunctor"


def short(name):
    n = name.replace("void ", "")
    if n.startswith("at::native::") or n.startswith("ck::") or n.startswith("_ZN"):
        return n[:110]
    return n.split("(")[0][:110]


def steps_of(trace_path):
    rows = list(csv.DictReader(open(trace_path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    steps, cur = [], []
    for i, r in enumerate(rows):
        cur.append(r)
        # the optimizer is several launches: the step ends with the last of the run
        if STEP_END in r["Kernel_Name"] and (i + 1 == len(rows) or
                                             STEP_END not in rows[i + 1]["Kernel_Name"]):
            steps.append(cur)
            cur = []
    return steps


def main():
    tag = sys.argv[1]
    keep = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    base = f"gpurun_out/{tag}_train/"
    steps = steps_of(base + "kt/p_kernel_trace.csv")[-keep:]
    n = len(steps)
    agg = collections.defaultdict(list)
    span = []
    for s in steps:
        span.append((int(s[-1]["End_Timestamp"]) - int(s[0]["Start_Timestamp"])) / 1e6)
        for r in s:
            agg[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    total = sum(sum(v) for v in agg.values())
    out = f"profiles/{tag}_train_kernel_stats.csv"
    with open(out, "w") as fo:
        w = csv.writer(fo)
        w.writerow(["kernel", "calls_per_step", "avg_us", "us_per_step", "percent"])
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([k, round(len(v) / n, 2), round(st.mean(v) / 1e3, 2),
                        round(sum(v) / n / 1e3, 1), round(100 * sum(v) / total, 2)])
        w.writerow(["# steady-state steps", n, "kernel time per step (ms)",
                    round(total / n / 1e6, 3), ""])
        w.writerow(["# step span first kernel start -> last kernel end (ms, under rocprofv3)",
                    round(st.median(span), 3), "", "", ""])
    print(out, f"{n} steps, {total / n / 1e6:.2f} ms of kernels per step, span {st.median(span):.2f} ms")

    # counter passes: the last step of each pass, keyed by (kernel, occurrence index in the step)
    def last_step_counters(d):
        rows = list(csv.DictReader(open(base + d + "/p_counter_collection.csv")))
        by_disp = collections.OrderedDict()
        for r in rows:
            by_disp.setdefault(int(r["Dispatch_Id"]), []).append(r)
        disp = sorted(by_disp)
        ends = [i for i, did in enumerate(disp) if STEP_END in by_disp[did][0]["Kernel_Name"] and
                (i + 1 == len(disp) or STEP_END not in by_disp[disp[i + 1]][0]["Kernel_Name"])]
        lo, hi = (ends[-2] + 1, ends[-1] + 1) if len(ends) > 1 else (0, len(disp))
        res = collections.defaultdict(lambda: collections.defaultdict(list))
        dur = collections.defaultdict(list)
        for did in disp[lo:hi]:
            rs = by_disp[did]
            k = short(rs[0]["Kernel_Name"])
            dur[k].append(int(rs[0]["End_Timestamp"]) - int(rs[0]["Start_Timestamp"]))
            for r in rs:
                res[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        return res, dur
    f, _ = last_step_counters("pmc_fetch")
    wr, _ = last_step_counters("pmc_write")
    s, du = last_step_counters("pmc_sq")
    rows = []
    for k in sorted(s, key=lambda k: -sum(du[k])):
        fe = st.mean(f[k]["FETCH_SIZE"]) * 1024 if k in f else 0
        wb = st.mean(wr[k]["WRITE_SIZE"]) * 1024 if k in wr else 0
        d = st.mean(du[k])
        sq = {c: st.mean(v) for c, v in s[k].items()}
        cyc = sq.get("GRBM_GUI_ACTIVE", 0) / 8
        wc = sq.get("SQ_WAVE_CYCLES", 0)
        hbm = 2 * fe + wb
        rows.append(dict(
            kernel=k, launches_per_step=len(du[k]), avg_us=round(d / 1e3, 2),
            us_per_step=round(sum(du[k]) / 1e3, 1),
            fetch_size_bytes=int(fe), write_size_bytes=int(wb), hbm_bytes_corrected=int(hbm),
            hbm_GBps=round(hbm / d, 1),
            mfma_busy_cycles=int(sq.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)),
            mfma_util=round(sq.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cyc * 1024), 4) if cyc else 0,
            wait_any_frac=round(sq.get("SQ_WAIT_ANY", 0) / wc, 3) if wc else 0,
            wait_inst_frac=round(sq.get("SQ_WAIT_INST_ANY", 0) / wc, 3) if wc else 0))
    out = f"profiles/{tag}_train_pmc_summary.csv"
    with open(out, "w") as fo:
        w = csv.DictWriter(fo, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)
    print(out)
    for r in rows[:30]:
        print(f"{r['kernel'][:58]:58s} x{r['launches_per_step']:4d} {r['avg_us']:8.2f} us "
              f"{r['us_per_step']:8.1f}/step {r['hbm_GBps']:7.1f} GB/s mfma {r['mfma_util']:.3f} "
              f"wait_inst {r['wait_inst_frac']}")


if __name__ == "__main__":
    main()
