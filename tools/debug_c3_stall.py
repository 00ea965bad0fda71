import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import bench
dev = torch.device("cuda", 0)
model, layers = bench.build_c3(dev)
x = torch.randn(1 << 20, 32, device=dev)
with torch.no_grad():
    for _ in range(10):
        model.log_prob(x, return_sum=True)
    torch.cuda.synchronize()
    for trial in range(3):
        model.layer_events = []
        t_host = []
        for s in range(30):
            t0 = time.perf_counter()
            model.log_prob(x, return_sum=True)
            t_host.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
        ev = model.layer_events; model.layer_events = None
        ms = [a.elapsed_time(b) for a, b in ev]
        worst = max(range(len(ms)), key=lambda i: ms[i])
        print(f"trial {trial}: worst layer time {ms[worst]*1e3:.0f} us at step {worst // 9} layer {worst % 9}; "
              f"median NSF {sorted(ms[0::3])[len(ms[0::3])//2]*1e3:.0f} us; max host enqueue {max(t_host)*1e3:.2f} ms at step {t_host.index(max(t_host))}; "
              f"mem reserved {torch.cuda.memory_reserved()/2**30:.2f} GiB, n_alloc_retries {torch.cuda.memory_stats().get('num_alloc_retries')}")
        big = [(i // 9, i % 9, round(v * 1e3)) for i, v in enumerate(ms) if v > 1.0]
        print("   layers > 1 ms:", big[:10])
