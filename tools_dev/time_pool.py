import sys, os
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/ssd_object_detection_amd.py') else os.getcwd())
import torch
import ssd_object_detection_amd.ops as ops
for (B,H,C,same) in [(64,300,64,False),(64,150,128,False),(64,75,256,True)]:
    x = torch.randn((B,H,H,C), device="cuda").relu().bfloat16()
    y, code = ops.maxpool2x2_fwd_argmax(x, same=same)
    dy = torch.randn_like(y)
    dx = ops.maxpool2x2_bwd_argmax(code, dy, x.shape)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.maxpool2x2_bwd_argmax(code, dy, x.shape, out=dx)
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)/10*1e3
    gb=(dy.numel()*2+code.numel()*4+dx.numel()*2)/1e9
    print(f"pool bwd {H} C{C}: {us:.1f} us  {gb/us*1e3:.2f} TB/s", flush=True)
