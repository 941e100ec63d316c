#!/usr/bin/env python
"""The per-kernel table of DESIGN.md section 3, printed from profiles/<tag>_kernel_rooflines.json (nothing typed by hand):
    python tools/kernel_table.py r06"""
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
k = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", f"{tag}_kernel_rooflines.json")))
print("| kernel | ms / solve | VALU / SALU instr per wave | fp64 issue fraction of the chip | waves parked (`SQ_WAIT_ANY`) | LDS bank conflicts per LDS instr | HBM (FETCH×1…×2 + WRITE) / 8 TB/s |")
print("|---|---|---|---|---|---|---|")
for name, e in list(k["kernels"].items())[:12]:
    hb = e.get("hbm_frac", [0, 0])
    print(f"| `{name}` | {e['ms_per_solve']:.2f} | {e['valu_per_wave']:.0f} / {e.get('salu_per_wave', 0):.0f} | {e['valu_issue_frac']:.2f} | "
          f"{e['wait_share']:.2f} | {e.get('lds_bank_conflict_per_lds_inst', 0):.2f} | {hb[0]:.2f} … {hb[1]:.2f} |")
print(f"\n(durations: {k['durations_from']}; one sequential solve of 65536 problems)")
