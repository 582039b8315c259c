#!/usr/bin/env python3
"""Register / spill / scratch table of every kernel in a hipcc -save-temps gfx950 assembly file (.s): the check that a kernel edit did
not push a 256-VGPR ping-pong kernel into scratch.  usage: tools/kernel_regs.py build/gemm_r8-hip-amdgcn-amd-amdhsa-gfx950.s"""
import re
import subprocess
import sys

s = open(sys.argv[1]).read()
for b in s.split("- .agpr_count:")[1:]:
    get = lambda k: re.search(r"\.%s:\s+(\S+)" % k, b).group(1)
    name = subprocess.run(["c++filt", get("name")], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"^void \(anonymous namespace\)::", "", name).replace("(insv2v_gemm_desc)", "")
    print(f"{name[:90]:90s} vgpr {get('vgpr_count'):>3s} agpr {b.splitlines()[0].strip():>3s} vspill {get('vgpr_spill_count'):>3s} "
          f"sgpr {get('sgpr_count'):>3s} sspill {get('sgpr_spill_count'):>3s} scratch {get('private_segment_fixed_size'):>4s} lds {get('group_segment_fixed_size')}")
