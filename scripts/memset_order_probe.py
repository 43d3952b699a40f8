"""Is hipMemset on device memory host-synchronous on this runtime?  A backlog on the NULL stream, then hipMemset of a
fresh allocation, then a copy on a new non-blocking stream: does the copy see the memset?"""
import ctypes, time, torch
hip = ctypes.CDLL("libamdhip64.so")
a = torch.randn(4096, 4096, device="cuda")
torch.cuda.synchronize()
for trial in range(4):
    st = torch.cuda.Stream()
    p = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(768)) == 0
    host = (ctypes.c_uint8 * 768)()
    torch.cuda.synchronize()
    for _ in range(30):
        a @ a
    t0 = time.perf_counter()
    assert hip.hipMemset(p, 0x55, ctypes.c_size_t(768)) == 0
    t1 = time.perf_counter()
    assert hip.hipMemcpyAsync(host, p, ctypes.c_size_t(768), 2, ctypes.c_void_p(st.cuda_stream)) == 0
    assert hip.hipStreamSynchronize(ctypes.c_void_p(st.cuda_stream)) == 0
    t2 = time.perf_counter()
    seen = sum(1 for b in host if b == 0x55)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print("trial %d: hipMemset returned after %.2f ms; copy on the new stream saw %d of 768 bytes set (%.2f ms later); backlog drained after %.2f ms"
          % (trial, (t1 - t0) * 1e3, seen, (t2 - t1) * 1e3, (t3 - t0) * 1e3))
