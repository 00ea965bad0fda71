"""profiles/rN/<workload>_pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of tools/profile_bench.sh.
usage: make_traffic_json.py <gpurun_out/prof_TAG> <workload> <kernel substring> <out.json> [launches per unit]"""
import csv, glob, json, os, sys
d, workload, kern, out = sys.argv[1:5]
mult = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0


def mean_of(sub, counter):
    f = sorted(glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True))[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
            if r["Counter_Name"] == counter and kern in r["Kernel_Name"]]
    return sum(vals) / len(vals), len(vals)


fetch, nf = mean_of("pmc_fetch", "FETCH_SIZE")
write, nw = mean_of("pmc_write", "WRITE_SIZE")
fetch, write = fetch * mult, write * mult
json.dump({
    "kernel": kern, "workload": workload, "launches_per_unit": mult, "FETCH_SIZE_KB": round(fetch, 1), "WRITE_SIZE_KB": round(write, 1),
    "dispatches": {"fetch_pass": nf, "write_pass": nw},
    "correction": "gfx950: FETCH_SIZE reports half of a wide coalesced streaming read (MI355X_MICROARCH.md, HBM): "
                  "traffic = (2*FETCH_SIZE + WRITE_SIZE) * 1024",
    "traffic_bytes_per_launch": (2 * fetch + write) * 1024,
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python3 bench.py --steps 20 --warmup 5 "
              f"--no-cpu-baseline --workload {workload}` (tools/profile_bench.sh)",
}, open(out, "w"), indent=1)
print(open(out).read())
