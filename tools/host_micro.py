import time, torch, ctypes
s1 = torch.cuda.Stream(); cur = torch.cuda.current_stream()
x = torch.zeros(1024, device="cuda")
def t(f, n=2000):
    for _ in range(50): f()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): f()
    dt=(time.perf_counter()-t0)/n*1e6; torch.cuda.synchronize(); return dt
print("current_stream()", t(lambda: torch.cuda.current_stream()))
print("raw current", t(lambda: torch._C._cuda_getCurrentRawStream(0)))
def ctxm():
    with torch.cuda.stream(s1): pass
print("with stream ctx", t(ctxm))
def setst():
    torch.cuda.set_stream(s1); torch.cuda.set_stream(cur)
print("set_stream x2", t(setst))
print("wait_stream", t(lambda: s1.wait_stream(cur)))
ev = torch.cuda.Event()
def evre():
    ev.record(cur); s1.wait_event(ev)
print("reused event record+wait", t(evre))
print("record_stream", t(lambda: x.record_stream(s1)))
print("torch.empty", t(lambda: torch.empty(4096, device="cuda")))
print("empty_like", t(lambda: torch.empty_like(x)))
class D(ctypes.Structure):
    _fields_=[(f"f{i}", ctypes.c_int32) for i in range(25)]
print("ctypes struct 25 kw", t(lambda: D(f0=1,f1=2,f2=3,f3=4,f4=5,f5=6,f6=7,f7=8,f8=9,f9=10,f10=1,f11=2,f12=3,f13=4,f14=5,f15=6,f16=7,f17=8,f18=9,f19=10,f20=1,f21=2,f22=3,f23=4,f24=5)))
d=D()
print("24-field tuple key", t(lambda: (d.f0,d.f1,d.f2,d.f3,d.f4,d.f5,d.f6,d.f7,d.f8,d.f9,d.f10,d.f11,d.f12,d.f13,d.f14,d.f15,d.f16,d.f17,d.f18,d.f19,d.f20,d.f21,d.f22,d.f23)))
print("data_ptr", t(lambda: x.data_ptr()))
print("c_void_p(data_ptr)", t(lambda: ctypes.c_void_p(x.data_ptr())))
