"""profiles/rNN_aux_dominant.json from the rocprofv3 --kernel-trace summaries of the aux configurations (benchmarks/rocpd_stats.py
markdown): per configuration the step's kernels by time, the dominant one first -- read by bench.py (aux.*.roofline.dominant_kernel).
    python benchmarks/aux_dominant.py out.json graphsage_maxpool=profiles/r05_maxpool_kernel_stats.md unsupervised=... rmat=... gcn=..."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SKIP = ("at::", "rocprim", "rocclr", "build_padded_table", "void  [", "unique_by_key", "Cijk", "hipcub")


def parse(path):
    rows = []
    for line in open(path):
        c = [x.strip() for x in line.strip().strip("|").split("|")]
        if len(c) < 7 or c[0] in ("kernel", "---") or c[0].startswith("-"):
            continue
        if any(s in c[0] for s in SKIP):
            continue
        try:
            rows.append((c[0], int(c[1]), float(c[2]), float(c[3])))
        except ValueError:
            continue
    return rows


def main():
    out_path = sys.argv[1]
    out = {"configs": {}}
    for spec in sys.argv[2:]:
        key, path = spec.split("=", 1)
        rows = parse(path)
        if not rows:
            continue
        steps = max(r[1] for r in rows)              # the step kernels run once per step: the largest call count
        step_rows = [r for r in rows if r[1] >= 0.5 * steps and "gather_mean_kernel" not in r[0]]
        step_rows.sort(key=lambda r: -r[2])
        tot = sum(r[2] for r in step_rows)
        out["configs"][key] = {"kernel": step_rows[0][0], "avg_us": step_rows[0][3], "share_of_step_kernel_time": step_rows[0][2] / tot,
                               "step_kernel_time_us": tot / steps,
                               "kernels": [{"kernel": r[0], "avg_us": r[3], "calls_per_step": round(r[1] / steps, 2)} for r in step_rows[:8]],
                               # the summary this was read from is committed under profiles/ with the same file name
                               "source": "profiles/" + os.path.basename(path)}
    p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "graphsage_amd", "_C", "build.stamp")
    out["lib_digest"] = open(p).read().strip() if os.path.exists(p) else None
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps({k: (v["kernel"], round(v["avg_us"], 1)) for k, v in out["configs"].items()}))


if __name__ == "__main__":
    main()
