"""Does the Infinity Cache keep the most recently WRITTEN part of a tensor larger than itself? Write S bytes front to back (a torch fill), then read 128 MB from the
front or from the back (torch sum) and time the read: GB/s of the read. If the tail is hot, a consumer that walks its tiles in REVERSE of its producer starts on hits."""
import torch, time
dev = torch.device("cuda", 0)
def t_read(x):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); s = x.sum(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)
R = 128 << 20
for S in (128 << 20, 256 << 20, 512 << 20, 1 << 30, 2 << 30):
    x = torch.empty(S // 4, dtype=torch.float32, device=dev)
    n = R // 4
    res = {}
    for where in ("front", "back", "front", "back"):
        best = 1e9
        for rep in range(5):
            x.fill_(1.0)
            torch.cuda.synchronize()
            v = x[:n] if where == "front" else x[-n:]
            best = min(best, t_read(v))
        res[where] = best
    print(f"tensor {S >> 20:5d} MB written front to back; then reading 128 MB: front {R / res['front'] / 1e6:7.1f} GB/s   back {R / res['back'] / 1e6:7.1f} GB/s", flush=True)
    del x
