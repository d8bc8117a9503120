"""Practical HBM ceiling of the box ("use the measured figure as the practical ceiling and report both"):
device-to-device copy and a read-only reduction over buffers far larger than the caches.  Prints one JSON line.
    python tools/hbm_copy_probe.py [GiB per buffer, default 4]"""
import json
import sys

import torch


def timed(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps / 1e3


def main():
    gib = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
    n = int(gib * (1 << 30)) // 8
    dev = torch.device("cuda", 0)
    x = torch.ones(n, dtype=torch.int64, device=dev)
    y = torch.empty_like(x)
    t_copy = timed(lambda: y.copy_(x), 10)
    t_read = timed(lambda: x.sum(), 10)
    t_fill = timed(lambda: y.fill_(3), 10)
    nbytes = n * 8
    print(json.dumps({"device": torch.cuda.get_device_name(0), "buffer_GiB": gib,
                      "copy_GBps_read_plus_write": 2 * nbytes / t_copy / 1e9, "read_GBps": nbytes / t_read / 1e9, "write_GBps": nbytes / t_fill / 1e9}))


if __name__ == "__main__":
    main()
