import ctypes, torch, time
hip = ctypes.CDLL("libamdhip64.so")
for nbytes in (512<<20, 1<<30):
    a = torch.empty(nbytes, dtype=torch.uint8, device="cuda"); b = torch.empty_like(a)
    s = torch.cuda.current_stream().cuda_stream
    for name, fn in (("hipMemcpyAsync", lambda: hip.hipMemcpyAsync(ctypes.c_void_p(b.data_ptr()), ctypes.c_void_p(a.data_ptr()), ctypes.c_size_t(nbytes), 3, ctypes.c_void_p(s))),
                     ("torch copy_", lambda: b.copy_(a))):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print(nbytes >> 20, "MiB", name, "%.3f ms" % (dt * 1e3), "%.2f TB/s (r+w)" % (2 * nbytes / dt / 1e12))
