import sys, os, torch
ROOT="/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "recommender-tensorflow_amd"))
from mi355x_rec import _lib
lib=_lib.load(); st=lambda: _lib.cur_stream()
n=65536*26
g=torch.Generator(device="cuda"); g.manual_seed(0)
for R in (26_000_000, 1<<21, 1<<20, 1<<14):
    rows=torch.randint(0,R,(n,),device="cuda",dtype=torch.int32,generator=g)
    se=torch.empty(n,dtype=torch.int32,device="cuda"); uq=torch.empty(n,dtype=torch.int32,device="cuda"); sg=torch.empty(n+1,dtype=torch.int32,device="cuda"); nu=torch.empty(1,dtype=torch.int32,device="cuda")
    ws=torch.empty(int(lib.mi_sort_unique_workspace_bytes(n))+256,dtype=torch.uint8,device="cuda")
    f=lambda: lib.mi_sort_unique_rows(rows.data_ptr(), n, R, se.data_ptr(), uq.data_ptr(), sg.data_ptr(), nu.data_ptr(), ws.data_ptr(), ws.numel(), st())
    for _ in range(5): f()
    torch.cuda.synchronize(); s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); s.record()
    for _ in range(30): f()
    e.record(); torch.cuda.synchronize()
    bits=(R-1).bit_length(); passes=(bits+8)//9; nb=(bits+passes-1)//passes
    print("R=%9d bits=%2d passes=%d x %d bits: %.1f us" % (R,bits,passes,nb,s.elapsed_time(e)/30*1e3))
