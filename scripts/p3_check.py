"""diagnostic: error of the product forms on same-signed all-ones-significand operands (bias) and on random operands"""
import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ait_amd import ops, _lib
def all_ones(shape, gen):
    e = torch.randint(-2, 3, shape, device="cuda", generator=gen).float()
    return (2.0 - 2.0 ** -23) * torch.exp2(e)
for (M, N, K, ta, sk) in [(33000, 512, 64, False, 1), (33000, 512, 512, False, 1), (33000, 512, 2048, False, 1), (512, 2048, 76800, True, 16)]:
    gen = torch.Generator(device="cuda").manual_seed(M + K)
    for kind in ("ones", "pos_random", "random"):
        if kind == "ones":
            a = all_ones((K, M) if ta else (M, K), gen); b = all_ones((K, N) if ta else (N, K), gen)
        else:
            a = torch.rand((K, M) if ta else (M, K), device="cuda", generator=gen) + 0.5
            b = torch.rand((K, N) if ta else (N, K), device="cuda", generator=gen) + 0.5
            if kind == "random":
                a = a * (torch.randint(0, 2, a.shape, device="cuda", generator=gen).float() * 2 - 1)
                b = b * (torch.randint(0, 2, b.shape, device="cuda", generator=gen).float() * 2 - 1)
        ref = (a.double().t() @ b.double()) if ta else (a.double() @ b.double().t())
        mag = (a.double().abs().t() @ b.double().abs()) if ta else (a.double().abs() @ b.double().abs().t())
        got = ops.gemm(a, b, trans_a=ta, trans_b=not ta, split_k=sk)
        _lib.NATIVE_F32 = True
        nat = ops.gemm(a, b, trans_a=ta, trans_b=not ta, split_k=sk)
        _lib.NATIVE_F32 = False
        es, en = (got.double() - ref) / mag, (nat.double() - ref) / mag
        line = "%-10s M=%d N=%d K=%d | split: min %.3g max %.3g mean %.3g | f32 instr: min %.3g max %.3g mean %.3g" % (
            kind, M, N, K, es.min(), es.max(), es.mean(), en.min(), en.max(), en.mean())
        if not ta and K >= 128:
            c = ops.gemm_p3(a, ops.p3_split(b))
            e3 = (c.double() - ref) / mag
            line += " | p3: min %.3g max %.3g mean %.3g" % (e3.min(), e3.max(), e3.mean())
        print(line)
