import torch, ctypes, subprocess
hip = ctypes.CDLL("libamdhip64.so")
a = torch.arange(1, 1025, dtype=torch.int32, device="cuda")
out = torch.zeros(256, dtype=torch.int32, device="cuda")
mod = ctypes.c_void_p(); fn = ctypes.c_void_p()
data = open("tools/probe/gk.co","rb").read()
assert hip.hipModuleLoadData(ctypes.byref(mod), data) == 0
assert hip.hipModuleGetFunction(ctypes.byref(fn), mod, b"k") == 0
args = (ctypes.c_void_p*3)(); 
pa = ctypes.c_void_p(a.data_ptr()); po = ctypes.c_void_p(out.data_ptr()); nb = ctypes.c_uint(4096)
arr = (ctypes.c_void_p*3)(ctypes.cast(ctypes.byref(pa), ctypes.c_void_p), ctypes.cast(ctypes.byref(po), ctypes.c_void_p), ctypes.cast(ctypes.byref(nb), ctypes.c_void_p))
assert hip.hipModuleLaunchKernel(fn, 1,1,1, 64,1,1, 0, None, arr, None) == 0
torch.cuda.synchronize()
print(out[:24].tolist())
