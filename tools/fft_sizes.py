import torch, time
for F in (168, 176, 180, 192, 200, 216, 224, 240, 256):
    x = torch.randn(20480, F, F, device="cuda")
    for _ in range(2):
        y = torch.fft.rfft2(x); z = torch.fft.irfft2(y, s=(F, F))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        y = torch.fft.rfft2(x); z = torch.fft.irfft2(y, s=(F, F))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print("F=%d  r2c+c2r of 20480 planes: %.2f ms   (%.1f GB/s of plane data)" % (F, dt * 1e3, 2 * x.numel() * 4 * 2 / dt / 1e9))
    del x, y, z
