"""The SQ-counter table of a profile (two rocprofv3 passes of eight counters: tools/profile.sh, tools/profile_config.sh):
where the wave cycles of every kernel go.  Used by summarize_profile.py and summarize_config_profile.py."""
import csv
import glob
import os

SQ_NAMES = ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
            "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM",
            "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAVES", "SQ_ACTIVE_INST_VMEM"]


def write_table(src, out_path, counter, first):
    """counter(pattern, name) -> (totals, durations_ms, calls) per kernel; first: the kernel launched once per batch."""
    if not glob.glob(os.path.join(src, "sq1/**/*_counter_collection.csv"), recursive=True):
        return False
    sq, sq_dur, nb_sq = {}, {}, 1
    for i, cname in enumerate(SQ_NAMES):
        tot, d, cl = counter(("sq1" if i < 8 else "sq2") + "/**/*_counter_collection.csv", cname)
        sq[cname] = tot
        if cname == "SQ_WAVE_CYCLES":
            sq_dur, nb_sq = d, max(cl[first], 1)
    with open(out_path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "ms_per_batch(pmc run)", "waves_per_batch", "wave_Mcycles_per_batch(quad-cycles x4)",
                    "frac_wait_any(s_waitcnt/barrier)", "frac_wait_inst_any(issue stall)", "frac_active_inst_any",
                    "frac_active_valu", "frac_active_lds", "frac_wait_inst_lds", "frac_active_vmem",
                    "mean_waves_in_flight(of 8192 slots)", "valu_insts_per_wave", "lds_insts_per_wave", "salu_insts_per_wave",
                    "vmem_insts_per_wave", "lds_bank_conflict_frac_of_lds_active"])
        for k in sorted(sq_dur, key=lambda k: -sq_dur[k]):
            if not k.startswith("k_"):
                continue
            wc = sq["SQ_WAVE_CYCLES"][k]
            if wc <= 0:
                continue
            waves = max(sq["SQ_WAVES"].get(k, 0.0), 1.0)
            ms = sq_dur[k] / nb_sq
            # SQ_WAVE_CYCLES and the WAIT / ACTIVE counters are in quad-cycles (MI355X_MICROARCH.md, cycle constants);
            # mean waves in flight = wave cycles / kernel cycles, with the kernel's cycles from its duration at 2.4 GHz
            in_flight = (wc * 4.0 / nb_sq) / (ms * 1e-3 * 2.4e9) if ms > 0 else 0.0
            fr = lambda n: f"{sq[n].get(k, 0.0) / wc:.3f}"  # noqa: E731
            lds_act = sq["SQ_LDS_IDX_ACTIVE"].get(k, 0.0)
            w.writerow([k, f"{ms:.4f}", f"{waves / nb_sq:.0f}", f"{wc * 4.0 / nb_sq / 1e6:.2f}", fr("SQ_WAIT_ANY"), fr("SQ_WAIT_INST_ANY"),
                        fr("SQ_ACTIVE_INST_ANY"), fr("SQ_ACTIVE_INST_VALU"), fr("SQ_ACTIVE_INST_LDS"), fr("SQ_WAIT_INST_LDS"),
                        fr("SQ_ACTIVE_INST_VMEM"), f"{in_flight:.0f}",
                        f"{sq['SQ_INSTS_VALU'].get(k, 0.0) / waves:.0f}", f"{sq['SQ_INSTS_LDS'].get(k, 0.0) / waves:.0f}",
                        f"{sq['SQ_INSTS_SALU'].get(k, 0.0) / waves:.0f}", f"{sq['SQ_INSTS_VMEM'].get(k, 0.0) / waves:.0f}",
                        f"{sq['SQ_LDS_BANK_CONFLICT'].get(k, 0.0) / lds_act:.3f}" if lds_act > 0 else ""])
    print(open(out_path).read())
    return True
