"""The N>1 path on CPU: two gloo ranks exercise ait_amd.distributed (init from the
torch.distributed.run environment, DDP wrap with static graph + a never-used parameter, pair
sharding, max-over-ranks timing) with a stand-in module (the HIP ops need a GPU)."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, torch, torch.nn as nn
    sys.path.insert(0, %r)
    from ait_amd import distributed as D
    rank, local_rank, world = D.init()
    assert world == 2 and rank == int(os.environ["RANK"])
    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.a = nn.Linear(8, 8); self.b = nn.Linear(8, 1)
            self.unused = nn.Linear(8, 8)        # like RCNN_base.backbone.fc: never gets a grad
        def forward(self, x):
            return self.b(torch.relu(self.a(x)))
    torch.manual_seed(0)
    net = Net()
    ddp = D.wrap(net, local_rank)
    clock = D.BucketClock(ddp)                   # (bench.py's note of when the buckets reach the all-reduce)
    lo, hi = D.shard_slice(6, rank, world)
    assert (lo, hi) == ((0, 3) if rank == 0 else (3, 6))
    torch.manual_seed(100)
    x = torch.randn(6, 8)[lo:hi]
    for _ in range(3):                          # static_graph needs >1 iteration to settle
        net.zero_grad()
        loss = ddp(x).sum()
        clock.start()
        loss.backward()
    s = clock.summary()
    assert s["buckets"] >= 1 and 0 <= s["first_ready_ms"] <= s["last_ready_ms"]
    g = net.a.weight.grad.clone()
    ref = [torch.zeros_like(g) for _ in range(world)]
    torch.distributed.all_gather(ref, g)
    assert torch.allclose(ref[0], ref[1]), "gradients were not all-reduced"
    assert net.unused.weight.grad is None
    # equals the mean over ranks of the local gradients
    net2 = Net(); net2.load_state_dict(net.state_dict())
    full = torch.randn(6, 8, generator=torch.Generator().manual_seed(100))
    torch.manual_seed(100); full = torch.randn(6, 8)
    net2(full).sum().backward()
    assert torch.allclose(g, net2.a.weight.grad / world, atol=1e-6)
    t = D.max_over_ranks(1.0 + rank, torch.device("cpu"))
    assert t == 2.0
    D.barrier()
    print("rank", rank, "ok")
""")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gloo_data_parallel(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1",
                   AIT_DIST_BACKEND="gloo", HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, o)
        assert "rank %d ok" % r in o


RS_AG_WORKER = textwrap.dedent("""
    import os, sys, torch, torch.nn as nn, torch.distributed as dist
    sys.path.insert(0, %r)
    from ait_amd import distributed as D
    rank, local_rank, world = D.init()
    # gloo has neither reduce_scatter_tensor nor all_gather_into_tensor: the hook's two collectives are emulated with what
    # it has (same results); everything else -- padding to a multiple of the world size, the average, the copy back into
    # the bucket, the future chain DDP waits on -- is the product's code
    def rs(shard, flat, group):
        tmp = flat.clone()
        fut = dist.all_reduce(tmp, group=group, async_op=True).get_future()
        return fut.then(lambda f: shard.copy_(tmp.view(world, -1)[rank]))
    def ag(flat, shard, group):
        parts = [torch.empty_like(shard) for _ in range(world)]
        fut = dist.all_gather(parts, shard, group=group, async_op=True).get_future()
        return fut.then(lambda f: flat.copy_(torch.cat(parts)))
    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.a = nn.Linear(7, 8); self.b = nn.Linear(8, 3)      # 56 + 8 + 24 + 3 = 91 values: an odd bucket, padded to 92
        def forward(self, x):
            return self.b(torch.tanh(self.a(x)))
    grads = {}
    for kind in ("allreduce", "rs_ag"):
        torch.manual_seed(0)
        net = Net()
        ddp = D.wrap(net, local_rank, bucket_mb=1)
        inner = D.make_exchange_hook(kind, rs, ag)
        clock = D.BucketClock(ddp, kind, inner=inner)
        torch.manual_seed(100 + rank)
        x = torch.randn(5, 7)
        for _ in range(3):
            net.zero_grad()
            clock.start()
            ddp(x).pow(2).sum().backward()
        assert clock.summary()["buckets"] >= 1 and clock.exchange == kind
        if kind == "rs_ag":                     # one padded copy + one shard per bucket, allocated once
            assert len(inner.cache) >= 1 and all(b[1] is not None and b[1].numel() == 92 and b[2].numel() == 46 for b in inner.cache.values())
        grads[kind] = torch.cat([p.grad.flatten() for p in net.parameters()])
    assert torch.allclose(grads["rs_ag"], grads["allreduce"], rtol=1e-6, atol=1e-7), (grads["rs_ag"] - grads["allreduce"]).abs().max()
    assert float(grads["rs_ag"].abs().max()) > 0
    try:
        D.make_exchange_hook("ring")
        raise SystemExit("an unknown exchange was accepted")
    except ValueError:
        pass
    D.barrier()
    print("rank", rank, "ok")
""")


def test_reduce_scatter_all_gather_exchange_equals_the_all_reduce(tmp_path):
    """the opt-in rs_ag bucket exchange (ait_amd.distributed.make_exchange_hook): same gradients as DDP's all-reduce hook on
    two gloo ranks, with a bucket whose size is not a multiple of the world size (the two collectives emulated: see the worker)"""
    script = tmp_path / "worker_rs_ag.py"
    script.write_text(RS_AG_WORKER % ROOT)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1",
                   AIT_DIST_BACKEND="gloo", HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, o)
        assert "rank %d ok" % r in o


def test_shard_slice_covers_everything():
    from ait_amd.distributed import shard_slice
    for n in (0, 1, 7, 64):
        for w in (1, 2, 3, 8):
            spans = [shard_slice(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def _bench(args, extra_env, drop=("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, timeout=300)


def test_bench_gpus_n_spawns_ranks_and_refuses_a_mismatch():
    """bench.py --gpus N on a GPU-less host: (1) without enough GPUs the parent refuses before starting anything;
    (2) with the gloo test hook it starts N children, each of which refuses to run without a GPU -- the parent reports
    the failure with a non-zero exit and no JSON line; (3) inside a rank environment whose WORLD_SIZE is not --gpus the
    rank refuses instead of printing a line for another job size."""
    r = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], {})
    assert r.returncode == 2 and "GPU(s) visible" in r.stderr and r.stdout.strip() == ""
    r = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"AIT_DIST_BACKEND": "gloo", "GLOO_SOCKET_IFNAME": "lo"})
    assert r.returncode == 1 and "rank exit codes" in r.stderr and "{" not in r.stdout
    r = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"})
    assert r.returncode != 0 and "refusing" in r.stderr


def test_spawn_ranks_cannot_hang(tmp_path, monkeypatch):
    """bench.spawn_ranks: a rank that ignores SIGTERM after another one failed is killed after the grace period; a job
    that never finishes is killed at its deadline (exit code 124); neither returns a JSON line"""
    import importlib
    import time
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    monkeypatch.setenv("AIT_DIST_BACKEND", "gloo")
    stubborn = tmp_path / "stubborn.py"
    stubborn.write_text(textwrap.dedent("""
        import os, signal, sys, time
        signal.signal(signal.SIGTERM, signal.SIG_IGN)          # like a rank stuck inside a collective
        if os.environ["RANK"] == "1" and sys.argv[1] == "fail":
            sys.exit(3)
        print('{"not": "a result"}', flush=True)
        time.sleep(600)
    """))
    t0 = time.monotonic()
    assert bench.spawn_ranks(2, ["fail"], deadline_s=120, grace_s=1.0, script=str(stubborn)) == 1
    assert time.monotonic() - t0 < 30
    t0 = time.monotonic()
    assert bench.spawn_ranks(2, ["sleep"], deadline_s=2.0, grace_s=1.0, script=str(stubborn)) == 124
    assert time.monotonic() - t0 < 30


def test_spawn_ranks_retries_when_the_port_was_taken(tmp_path, monkeypatch, capfd):
    import importlib
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    monkeypatch.setenv("AIT_DIST_BACKEND", "gloo")
    flag = tmp_path / "second_try"
    script = tmp_path / "once_busy.py"
    script.write_text(textwrap.dedent("""
        import os, sys
        flag = %r
        if not os.path.exists(flag):
            if os.environ["RANK"] == "0":
                open(flag, "w").close()
                sys.stderr.write("RuntimeError: The server socket has failed to listen ... EADDRINUSE: address already in use\\n")
                sys.exit(1)
            import time; time.sleep(30)
        if os.environ["RANK"] == "0":
            print('{"ok": %%s}' %% os.environ["MASTER_PORT"])
    """ % str(flag)))
    assert bench.spawn_ranks(2, [], deadline_s=60, grace_s=1.0, script=str(script)) == 0
    out = capfd.readouterr()
    assert '{"ok": ' in out.out and "starting the ranks again" in out.err


def _fake_sysfs(root, gpus):
    """gpus: list of (domain, bus, numa_node, cpulist); KFD node 0 is the CPU"""
    top = root / "class" / "kfd" / "kfd" / "topology" / "nodes"
    (top / "0").mkdir(parents=True)
    (top / "0" / "properties").write_text("cpu_cores_count 64\nsimd_count 0\nlocation_id 0\n")
    for i, (dom, bus, node, cpus) in enumerate(gpus):
        d = top / str(i + 1)
        d.mkdir()
        d.joinpath("properties").write_text("cpu_cores_count 0\nsimd_count 1024\nlocation_id %d\ndomain %d\n" % (bus << 8, dom))
        p = root / "bus" / "pci" / "devices" / ("%04x:%02x:00.0" % (dom, bus))
        p.mkdir(parents=True)
        p.joinpath("numa_node").write_text("%d\n" % node)
        p.joinpath("local_cpulist").write_text(cpus + "\n")


def test_rank_binding_follows_the_gpus_numa_node(tmp_path, monkeypatch):
    """ait_amd.distributed.bind_rank_to_gpu_numa: sysfs only (KFD topology -> PCI function -> numa_node / local_cpulist);
    ranks sharing a node split its cores; visible-device lists are honoured; a silent sysfs leaves the process alone"""
    import os as _os
    from ait_amd import distributed as D
    for v in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(v, raising=False)
    _fake_sysfs(tmp_path, [(0, 0x05, 0, "0-3"), (0, 0x15, 0, "0-3"), (0, 0x85, 1, "4-7"), (0, 0x95, 1, "4-7")])
    monkeypatch.setattr(_os, "sched_getaffinity", lambda pid: set(range(8)))
    assert D.gpu_numa_cpus(2, str(tmp_path)) == (1, {4, 5, 6, 7})
    got = []
    r = D.bind_rank_to_gpu_numa(3, 4, str(tmp_path), setaffinity=got.append)
    assert r["bound"] and r["numa_node"] == 1 and r["ranks_on_node"] == 2 and got == [{6, 7}]
    got = []
    r = D.bind_rank_to_gpu_numa(0, 1, str(tmp_path), setaffinity=got.append)
    assert r["bound"] and got == [{0, 1, 2, 3}]
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "2,3")
    assert D.gpu_numa_cpus(0, str(tmp_path)) == (1, {4, 5, 6, 7})
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "GPU-abcdef")
    assert D.gpu_numa_cpus(0, str(tmp_path)) == (None, None)
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    got = []
    r = D.bind_rank_to_gpu_numa(0, 1, str(tmp_path / "nothing_here"), setaffinity=got.append)
    assert not r["bound"] and got == []
    monkeypatch.setattr(_os, "sched_getaffinity", lambda pid: {0, 1})          # a cgroup that excludes the GPU's node
    r = D.bind_rank_to_gpu_numa(2, 4, str(tmp_path), setaffinity=got.append)
    assert not r["bound"] and got == []
