#!/bin/bash
# PMF pass with labels: parity tests, then the kernel at 100M x 150 with three clusters - label bytes against 2-bit labels, 4 / 6 groups
out=gpurun_out/${1:-pmf}
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py -x -q -k "pmf" > $out/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 $out/pytest.log
python3 - > $out/pmf3.txt 2>&1 <<'PY'
import os, torch
from qvz_amd import kernels as K
n, c, k = 100_000_000, 150, 3
sym = K.qgen1(3, n, c)
lab = (torch.arange(sym.shape[1], device="cuda") * 2654435761 >> 7).remainder(3).to(torch.uint8)
def run(tag, **env):
    for key in ("QVZ_PMF_LABELS", "QVZ_PMF_Q"):
        os.environ.pop(key, None)
    os.environ.update(env)
    cnt = torch.zeros((k, c, 72, 72), dtype=torch.int32, device="cuda")
    K.pmf_hist(sym, n, labels=lab, k=k, cnt=cnt)
    ms = []
    for _ in range(5):
        cnt.zero_()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); K.pmf_hist(sym, n, labels=lab, k=k, cnt=cnt); b.record(); torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
    print(f"{tag:32s} {min(ms):.3f} ms  {(n * c + n) / min(ms) / 1e6:.0f} GB/s = {(n * c + n) / min(ms) / 1e6 / 8000:.3f} of 8 TB/s", flush=True)
    return cnt
a = run("label bytes, 4 groups", QVZ_PMF_LABELS="bytes")
same = True
for g, r in ((3, 2), (2, 3), (0, 0)):
    for q in (4, 6):
        os.environ["QVZ_PMF_G"], os.environ["QVZ_PMF_R"] = str(g), str(r)
        x = run(f"2-bit labels, Q={q} G={g} R={r}", QVZ_PMF_Q=str(q))
        same = same and bool(torch.equal(a, x))
print("equal:", same)
for key in ("QVZ_PMF_LABELS", "QVZ_PMF_Q", "QVZ_PMF_G", "QVZ_PMF_R"):
    os.environ.pop(key, None)
cnt1 = torch.zeros((1, c, 72, 72), dtype=torch.int32, device="cuda")
for _ in range(4):
    cnt1.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); K.pmf_hist(sym, n, cnt=cnt1); e1.record(); torch.cuda.synchronize()
    print(f"one cluster: {e0.elapsed_time(e1):.3f} ms (the call: alphabet sample + kernel)")
PY
cat $out/pmf3.txt
