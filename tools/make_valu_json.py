"""profiles/rN/<workload>_valu_issue.json from the SQ counter passes of tools/profile_bench.sh: how much of the chip's
vector-instruction issue capacity the dominant kernel used (the C3 spline kernel is VALU / transcendental bound by
construction).
usage: make_valu_json.py <gpurun_out/prof_TAG> <workload> <kernel substring> <out.json> [launches per unit]
(launches per unit: the substring matches that many kernels whose counters are averaged per dispatch and then summed --
c3t's gradient pass is two stage kernels per layer)"""
import csv, glob, json, os, sys
d, workload, kern, out = sys.argv[1:5]
mult = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0


def mean_of(sub, counter):
    for f in sorted(glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True)):
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
                if r["Counter_Name"] == counter and kern in r["Kernel_Name"]]
        if vals:
            return sum(vals) / len(vals)
    return None


def kernel_ns_mean():
    """mean over the matching kernels of their average duration (each weighted by its calls)"""
    for f in sorted(glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True)):
        tot = calls = 0.0
        for r in csv.DictReader(open(f)):
            if kern in r["Name"]:
                tot += float(r["AverageNs"]) * float(r["Calls"])
                calls += float(r["Calls"])
        if calls:
            return tot / calls
    return None


def scaled(v):
    return None if v is None else v * mult


valu = scaled(mean_of("pmc_sq", "SQ_INSTS_VALU"))            # wave-level vector instructions, MFMAs included
mfma = scaled(mean_of("pmc_sq2", "SQ_INSTS_MFMA"))
mfma_busy = scaled(mean_of("pmc_sq", "SQ_VALU_MFMA_BUSY_CYCLES"))
gui = scaled(mean_of("pmc_sq2", "GRBM_GUI_ACTIVE"))          # summed over the 8 XCDs
wave_cycles = scaled(mean_of("pmc_sq", "SQ_WAVE_CYCLES"))
active = scaled(mean_of("pmc_sq", "SQ_ACTIVE_INST_ANY"))
active_valu = scaled(mean_of("pmc_sq2", "SQ_ACTIVE_INST_VALU"))  # quad-cycles the waves spent inside vector instructions
ns = scaled(kernel_ns_mean())
cycles = gui / 8.0                                     # shader cycles of the dispatch
simds = 256 * 4
# MI355X_MICROARCH.md: a SIMD issues one wave64 vector instruction per 2 cycles (32 lanes / cycle); an MFMA
# (16x16x16 / 16x16x32 f16) holds the SIMD's vector issue for 8 cycles
plain = valu - mfma
issue_cycles = 2.0 * plain + 8.0 * mfma
json.dump({
    "kernel": kern, "workload": workload, "launches_per_unit": mult,
    "SQ_INSTS_VALU": valu, "SQ_INSTS_MFMA": mfma, "SQ_VALU_MFMA_BUSY_CYCLES": mfma_busy, "GRBM_GUI_ACTIVE": gui,
    "SQ_WAVE_CYCLES": wave_cycles, "SQ_ACTIVE_INST_ANY": active, "kernel_avg_ns_profiled": ns,
    "effective_clock_GHz": cycles / ns if ns else None,
    "valu_issue_peak_instr_per_cycle": simds / 2.0,
    "valu_instr_per_cycle": valu / cycles,
    "valu_issue_frac": issue_cycles / (simds * cycles),
    "mfma_pipe_frac": mfma_busy / (simds * cycles) if mfma_busy else None,
    # measured, not priced: the share of the waves' resident cycles spent INSIDE a vector instruction (both counters are
    # quad-cycles summed over the waves), the average length of one, and what that makes of the SIMDs' time
    "SQ_ACTIVE_INST_VALU": active_valu,
    "wave_cycles_in_valu_frac": active_valu / wave_cycles if active_valu and wave_cycles else None,
    "cycles_per_valu_instr": 4.0 * active_valu / valu if active_valu else None,
    "simd_valu_busy_frac": 4.0 * active_valu / (simds * cycles) if active_valu else None,
    "definition": "valu_issue_frac = (2 cycles x (SQ_INSTS_VALU - SQ_INSTS_MFMA) + 8 cycles x SQ_INSTS_MFMA) / (1024 SIMDs x "
                  "GRBM_GUI_ACTIVE / 8): the share of the dispatch's SIMD cycles in which a vector instruction was being "
                  "issued at the hardware's peak issue rate (transcendentals priced like plain instructions: a lower "
                  "bound of the issue time).  simd_valu_busy_frac = 4 x SQ_ACTIVE_INST_VALU / the same SIMD cycles: how "
                  "long the SIMDs' vector units were actually occupied (one wave's instruction occupies its SIMD for "
                  "cycles_per_valu_instr cycles; MI355X_MICROARCH.md prices v_fma_f32 at 4 cycles for one wave's stream, "
                  "transcendentals at 8)",
    "source": "rocprofv3 --pmc passes of `python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload "
              f"{workload}` (tools/profile_bench.sh)",
}, open(out, "w"), indent=1)
print(open(out).read())
