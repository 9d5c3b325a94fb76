#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r05j; mkdir -p $o
for v in 0 21 22; do echo -n "lib ring$v (0 real, 21 no validity selects, 22 + no row shifts): "; SIMHAND_LIB=scripts/abl/libring$v.so timeout 200 python scripts/ring_abl.py 2>&1 | tail -1; done | tee $o/ring_abl2.txt
cat > /tmp/ring1.py <<'PY'
import sys, torch
sys.path.insert(0, ".")
from simhand_amd import ops, _lib
N, h, cin, cout = 2048, 14, 256, 256
d = ops.conv_desc(N, h, h, cin, cout, 3, 3, 1, 1, torch.bfloat16)
x = torch.zeros(N, h, h, cin, device="cuda", dtype=torch.bfloat16)
wk = ops.pack_krsc(torch.zeros(cout, cin, 3, 3, device="cuda"), torch.bfloat16)
for ring in (0, 1):
    _lib.load().simhand_test_switch(16, ring)
    for _ in range(3): ops.conv2d_fwd(d, x, wk, True)
torch.cuda.synchronize()
PY
SIMHAND_LIB=scripts/abl/libring0.so timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d /tmp/rc -o p -- python /tmp/ring1.py > /tmp/rc.log 2>&1 || tail -3 /tmp/rc.log
python scripts/pmc_dump.py /tmp/rc/p_results.db igemm256 | tee $o/ring_pmc.txt
