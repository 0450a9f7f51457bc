"""What HBM sustains on this box for the stream mixes the step's kernels have: read-only, write-only,
1:1 copy and 2:1 read/write (torch elementwise kernels on 2.56 GB tensors, i.e. N=10M x 64 fp32)."""
import json
import torch


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


def main():
    n = 10_000_000 * 64
    x = torch.rand(n, device="cuda")
    y = torch.rand(n, device="cuda")
    z = torch.empty(n, device="cuda")
    gb = n * 4 / 1e9
    out = {
        "bytes_per_tensor_GB": gb,
        "read_only (sum)": gb / timed(lambda: x.sum()),
        "write_only (fill)": gb / timed(lambda: z.zero_()),
        "copy 1r:1w": 2 * gb / timed(lambda: z.copy_(x)),
        "add 2r:1w": 3 * gb / timed(lambda: torch.add(x, y, out=z)),
    }
    print(json.dumps({k: (round(v, 1) if isinstance(v, float) else v) for k, v in out.items()}))


if __name__ == "__main__":
    main()
