import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
dev = torch.device("cuda:0")
m, _, _ = bench.build_model(bench.WORKLOADS["c3"], 0, dev)
opt = torch.optim.Adam(m.parameters(), lr=0.01)
def sync(): torch.cuda.synchronize()
tl = tb = ts = ti = 0.0
for it in range(8):
    sync(); t0 = time.perf_counter()
    opt.zero_grad()
    loss = m.loss()
    sync(); t1 = time.perf_counter()
    loss.backward()
    sync(); t2 = time.perf_counter()
    opt.step()
    sync(); t3 = time.perf_counter()
    v = loss.item()
    t4 = time.perf_counter()
    if it >= 3:
        tl += t1 - t0; tb += t2 - t1; ts += t3 - t2; ti += t4 - t3
k = 5
print("per step: loss %.1f ms, backward %.1f ms, opt.step %.2f ms, item %.2f ms, total %.1f ms" % (tl/k*1e3, tb/k*1e3, ts/k*1e3, ti/k*1e3, (tl+tb+ts+ti)/k*1e3))
# the same without the syncs in between
sync(); t0 = time.perf_counter()
for it in range(5):
    opt.zero_grad(); loss = m.loss(); loss.backward(); opt.step(); v = loss.item()
sync(); print("free-running: %.1f ms per step" % ((time.perf_counter() - t0) / 5 * 1e3))
