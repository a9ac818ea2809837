"""Shader clock / socket power sampled with rocm-smi while one kernel type runs back to back for a few
seconds: is the fp32 GEMM clock- or power-limited compared with a register-only MFMA loop?"""
import os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ait_amd import ops
stop = False
samples = []
def sampler():
    while not stop:
        out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
        m = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", out)
        p = re.search(r"Power \(W\): ([\d.]+)", out)
        if m and p:
            samples.append((int(m.group(1)), float(p.group(1))))
def run(name, fn, flops_per_call, seconds=5.0):
    global stop, samples
    fn(); torch.cuda.synchronize()
    stop, samples = False, []
    th = threading.Thread(target=sampler); th.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < seconds:
        for _ in range(50): fn()
        torch.cuda.synchronize(); n += 50
    dt = time.time() - t0
    stop = True; th.join()
    s = sorted(samples)[len(samples) // 4:] or [(0, 0)]       # drop the ramp-up quarter
    print("%-34s %6.1f TFLOP/s sustained  sclk median %4d MHz (max %d)  power median %3.0f W  [%d samples]"
          % (name, n * flops_per_call / dt / 1e12, s[len(s) // 2][0], max(x[0] for x in samples), sorted(x[1] for x in s)[len(s) // 2], len(samples)))
M = 76800
for (nm, m, n, k, ta, tb, sk) in [("ffn1 NT 76800x2048x512", M, 2048, 512, False, True, 1), ("ffn2 NT 76800x512x2048", M, 512, 2048, False, True, 1),
                                  ("wgrad TN 2048x512x76800", 2048, 512, M, True, False, 16)]:
    a = torch.randn((k, m) if ta else (m, k), device="cuda"); b = torch.randn((n, k) if tb else (k, n), device="cuda")
    out = torch.zeros(m, n, device="cuda")
    run("ait_gemm_f32 " + nm, lambda: ops.gemm(a, b, trans_a=ta, trans_b=tb, out=out, split_k=sk), 2.0 * m * n * k)
    if not ta:
        run("rocBLAS      " + nm, lambda: torch.mm(a, b.t() if tb else b, out=out), 2.0 * m * n * k)
