import torch, sys
sys.path.insert(0, '/root/repo')
from ait_amd import ops
for (M,N,K) in [(76800,1536,512),(33000,512,2048),(58800,2048,512)]:
    for spread in (0.0, 2.0):
        torch.manual_seed(M+N+K)
        a = torch.randn(M, K, device="cuda") * torch.exp(spread * torch.randn(1, K, device="cuda"))
        w = torch.randn(N, K, device="cuda") * torch.exp(spread * torch.randn(1, K, device="cuda"))
        wp = ops.p3_split(w)
        ref = a.double() @ w.double().t()
        mag = a.double().abs() @ w.double().abs().t()
        c = ops.gemm_p3(a, wp); raw = ops.gemm(a, w)
        ep = ((c.double()-ref).abs()/mag); er = ((raw.double()-ref).abs()/mag)
        print(M,N,K,spread, "p3 max %.3g rms %.3g | raw max %.3g rms %.3g | mean signed p3 %.3g raw %.3g" % (ep.max(), ep.square().mean().sqrt(), er.max(), er.square().mean().sqrt(), ((c.double()-ref)/mag).mean(), ((raw.double()-ref)/mag).mean()))
        # where is the max
        i = int(ep.argmax()); print("   argmax row %d col %d" % (i // N, i % N))
