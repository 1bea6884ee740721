import time, torch
n = 1 << 30
src_pin = torch.empty(n, dtype=torch.uint8).pin_memory()
src_pag = torch.empty(n, dtype=torch.uint8)
dst = torch.empty(n, dtype=torch.uint8, device="cuda")
for name, src in (("pinned", src_pin), ("pageable", src_pag)):
    dst.copy_(src, non_blocking=True); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(4):
        dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    print(name, "H2D GB/s", 4 * n / (time.perf_counter() - t) / 1e9, flush=True)
t = time.perf_counter(); x = torch.empty(n, dtype=torch.uint8).pin_memory(); print("pin_memory alloc 1GB s", time.perf_counter() - t)
import ctypes
t = time.perf_counter(); rc = torch.cuda.cudart().cudaHostRegister(src_pag.data_ptr(), n, 0); print("hostRegister 1GB s", time.perf_counter() - t, rc)
