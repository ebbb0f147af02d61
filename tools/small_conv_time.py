import torch, time
from dreammat_amd import hipops
dev='cuda'
for dt in (torch.bfloat16, torch.float16):
    x=torch.randn(8,512,512,128,device=dev).to(dt); w=(torch.randn(4,9*128,device=dev)*0.05).to(dt)
    for _ in range(3): y=hipops.conv3x3_small_nhwc(x,w,None,1,(1,1),0)
    torch.cuda.synchronize(); t=time.time()
    for _ in range(20): y=hipops.conv3x3_small_nhwc(x,w,None,1,(1,1),0)
    torch.cuda.synchronize(); print(dt, (time.time()-t)/20*1e6,'us')
