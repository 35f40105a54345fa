#!/usr/bin/env python3
"""CU-time utilisation per kernel from the committed counter passes (profiles/<dir>/pmc_summary.csv): the share of
(256 CUs x kernel duration) during which a workgroup of the kernel was resident,

    SQ_WAVE_CYCLES (quad-cycles summed over all waves) x 4 / (waves per workgroup x 256 CUs x kernel cycles),

kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs.  One-workgroup-per-CU kernels only (the table kernels and the decoded layers: LDS or
registers admit one): what is missing to 1.0 is round quantisation + dispatch tail — CUs waiting for the slowest workgroup —, as
opposed to what the resident waves do with their cycles (SQ_ACTIVE_INST_ANY / SQ_WAIT_INST_ANY / SQ_WAIT_ANY shares, same file).
usage: cu_time.py profiles/r5_v11 > profiles/r5_v11/cu_time_utilisation.csv"""
import csv
import sys


def main(d):
    rows = list(csv.DictReader(open(d + "/pmc_summary.csv")))
    print("kernel,waves_per_workgroup,kernel_cycles,cu_time_utilisation,issuing,wait_inst_any,wait_any")
    for r in rows:
        k = r["kernel"]
        if not k.startswith(("k_conv_aprx", "k_conv_sym", "k_conv_half8", "k_conv_dec", "k_fc_sym8", "k_fc_aprx")):
            continue
        try:
            wc, ga = float(r["SQ_WAVE_CYCLES"]), float(r["GRBM_GUI_ACTIVE"])
        except ValueError:
            continue
        waves = 8 if (k.startswith(("k_conv_sym8", "k_conv_half8", "k_fc_sym8", "k_conv_dec_nchw"))) else 16
        cyc = ga / 8.0
        print("%s,%d,%.4g,%.3f,%.2f,%.2f,%.2f" % (k.replace(",", "."), waves, cyc, wc * 4.0 / (cyc * waves * 256.0),
                                                  float(r["SQ_ACTIVE_INST_ANY"]) / wc, float(r["SQ_WAIT_INST_ANY"]) / wc,
                                                  float(r["SQ_WAIT_ANY"]) / wc))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "profiles/r5_v11")
