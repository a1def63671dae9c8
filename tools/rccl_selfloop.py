#!/usr/bin/env python3
"""RCCL point-to-point through TorchComm.halo on ONE GPU: world size 1, the rank sends to and receives from itself inside
one batch_isend_irecv group -- the only way to run the transport's code path without a second GPU."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")

import torch                               # noqa: E402
import torch.distributed as dist           # noqa: E402


def main():
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
    from ngsamg_amd import dist as D
    comm = D.TorchComm()
    n = 46225
    vec = torch.arange(3 * n, dtype=torch.float64, device="cuda")
    send = vec[:n].clone() * 2.0
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for rep in range(5):
            comm.halo([{0: send}], [{0: vec[2 * n:3 * n]}])       # into a slice of a larger buffer, like the ghost segment
        s.synchronize()
    ok = bool(torch.equal(vec[2 * n:], send)) and bool(torch.equal(vec[:2 * n], torch.arange(2 * n, dtype=torch.float64, device="cuda")))
    t0 = torch.cuda.Event(enable_timing=True)
    t1 = torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(s):
        t0.record()
        for rep in range(200):
            comm.halo([{0: send}], [{0: vec[2 * n:3 * n]}])
        t1.record()
        s.synchronize()
    print(f"self-loop halo of {n} doubles through RCCL: correct = {ok}, {t0.elapsed_time(t1) / 200 * 1e3:.1f} us per exchange")
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
