import ctypes, torch, os
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bufprobe.so"))
x = torch.arange(1024, dtype=torch.float32, device="cuda")
o = torch.zeros(256, device="cuda")
for nbytes, scale in ((4096, 16), (4096, 4), (512, 16), (4096, 64)):
    o.zero_()
    lib.probe(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(o.data_ptr()), nbytes, scale, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    print("nbytes", nbytes, "scale", scale, o[:12].tolist(), "... lane 33:", o[132:136].tolist())
