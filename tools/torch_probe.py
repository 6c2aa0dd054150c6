import os, sys, ctypes
order = sys.argv[1]
lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "py-graph-imputation_amd", "libgrim_hip.so")
if order == "torch_first":
    import torch
    print("torch sees", torch.cuda.is_available(), torch.cuda.device_count())
    if torch.cuda.is_available():
        torch.cuda.set_device(0); x = torch.ones(4, device="cuda"); print("torch tensor ok", float(x.sum()))
    L = ctypes.CDLL(lib); L.grim_device_count.restype = ctypes.c_int
    print("grim sees", L.grim_device_count())
else:
    L = ctypes.CDLL(lib); L.grim_device_count.restype = ctypes.c_int
    print("grim sees", L.grim_device_count())
    import torch
    print("torch sees", torch.cuda.is_available(), torch.cuda.device_count())
os.system("grep -E 'libamdhip64|libhsa-runtime' /proc/%d/maps | awk '{print $6}' | sort -u" % os.getpid())
