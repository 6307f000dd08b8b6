import importlib, sys, time, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch
pkg=importlib.import_module("visual-odometry-gpu_amd")
import oracle_lib as O
img=O.load_kitti(0)
p=pkg.default_params("gpu", nfeatures=1000, max_width=1241, max_height=376, max_batch=1, blur_levels=2)
c=pkg.Context(p)
for i in range(10): c.detect_and_compute(img)
c.enable_stage_timing(1)
acc={}
for i in range(50):
    c.detect_and_compute(img)
    for k,v in c.last_stage_times().items(): acc[k]=acc.get(k,0)+v/50
c.enable_stage_timing(0)
print({k:round(v*1000,1) for k,v in acc.items()})
import ctypes as C
t=time.perf_counter()
for i in range(200): c.detect_and_compute(img)
print("python call us", (time.perf_counter()-t)/200*1e6)
# raw C call without python allocations
lib=c._lib; h,w=img.shape; cap=995
kps=np.zeros((cap,2),np.int32); ang=np.zeros(cap,np.float32); desc=np.zeros((cap,32),np.uint8); cnt=C.c_int(0)
P=lambda a:a.ctypes.data_as(C.c_void_p)
t=time.perf_counter()
for i in range(200): lib.orbx_detect_and_compute(c._h,P(img),w,h,w,P(kps),P(ang),P(desc),None,None,None,cap,C.byref(cnt))
print("raw C call us", (time.perf_counter()-t)/200*1e6, cnt.value)
# device-resident single frame
d=torch.from_numpy(img).cuda(); torch.cuda.synchronize()
t=time.perf_counter()
for i in range(200):
    c.batch_device(d.data_ptr(),1,w,h); c.wait()
print("device-resident batch-1 us", (time.perf_counter()-t)/200*1e6)
