"""Aggregate a timed-region kernel stats csv (scripts/trace_stats.py) into categories, ms/step.
usage: trace_categories.py <stats.csv> <steps>"""
import collections, csv, sys
steps = float(sys.argv[2])
rows = list(csv.reader(l for l in open(sys.argv[1]) if not l.startswith('#')))[1:]
def c(n):
    if 'gemm_f32_' in n or 'gemm_bf16_kernel' in n: return 'ait_gemm'
    if 'Sp3AsmConv' in n: return 'miopen winograd'
    if n.startswith('igemm_'): return 'miopen ' + n[:9]
    if 'bwd_weight' in n or n.startswith('_ZN2ck'): return 'miopen ck'
    if n.startswith('Cijk'): return 'rocblas (miopen gemm convs + torch linear)'
    if 'batched_transpose' in n or 'transpose_' in n: return 'miopen layout transposes'
    if 'SubTensorOp' in n or 'Op1dTensor' in n or 'OpTensor' in n: return 'miopen tensor ops'
    if 'naive_conv' in n or 'miopen' in n.lower() or 'gridwise' in n.lower(): return 'miopen other'
    if 'at::native' in n or 'rocprim' in n or 'at::cuda' in n: return 'torch elementwise / reduce / index'
    if 'anonymous' in n:
        if 'gemm_bf16s_tn' in n or 'tn_reduce' in n: return 'ait_gemm bf16 storage: weight gradients (+ partial-tile reduce)'
        if 'gemm_bf16s' in n: return 'ait_gemm bf16 storage'
        if 'to_bf16' in n: return 'ait f32 -> bf16 conversions'
        if 'mha_core_bwd' in n: return 'ait fused attention block, backward (fc dgrad + selective heads + tiles)'
        if 'mha_core' in n: return 'ait fused attention block (fwd: tiles + selective heads + fc + LayerNorm)'
        for k in ('roi_align', 'attn', 'bn_act', 'ln_', 'sh_', 'nms', 'sk_', 'colsum', 'rep_sum'):
            if k in n: return 'ait ' + k.strip('_')
        return 'ait other'
    return 'other: ' + n[:50]
cat = collections.defaultdict(float)
for r in rows:
    cat[c(r[0])] += float(r[2]) / steps / 1e6
tot = sum(cat.values())
for k, v in sorted(cat.items(), key=lambda kv: -kv[1]):
    print("%8.3f ms  %5.1f %%  %s" % (v, 100 * v / tot, k))
print("%8.3f ms  total kernel time per step" % tot)
