"""Which torch streams share a hardware queue?  A device-side sleep on stream i, then a tiny kernel on stream j: if j's kernel
completes only after the sleep, i and j share a queue.     python scripts/exp_queues2.py [rccl]"""
import os, sys, time, torch
if len(sys.argv) > 1 and sys.argv[1] == "rccl":
    os.environ.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29549")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
main = torch.cuda.current_stream()
pool = [torch.cuda.Stream(device=dev) for _ in range(8)]
names = ["main"] + [f"s{i}" for i in range(8)]
streams = [main] + pool
x = torch.zeros(64, device=dev)
for s in streams:                       # first use of every stream
    with torch.cuda.stream(s): x.add_(1)
torch.cuda.synchronize()
SLEEP = 2_000_000                       # ~1 ms of device-side spinning
def blocked(i, j):
    torch.cuda.synchronize()
    with torch.cuda.stream(streams[i]): torch.cuda._sleep(SLEEP)
    t0 = time.perf_counter()
    with torch.cuda.stream(streams[j]): x.add_(1)
    streams[j].synchronize()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    return dt
print("rows: sleeping stream; columns: stream of the tiny kernel; * = waited for the sleep")
print("      " + " ".join(f"{n:>5s}" for n in names))
for i in range(len(streams)):
    row = []
    for j in range(len(streams)):
        if i == j: row.append("    -"); continue
        dt = min(blocked(i, j) for _ in range(2))
        row.append(f"{dt*1e3:4.2f}*" if dt > 0.4e-3 else f"{dt*1e3:5.2f}")
    print(f"{names[i]:>5s} " + " ".join(row), flush=True)
