import torch, time
a=torch.randn(4096,4096,device='cuda'); b=torch.randn(4096,4096,device='cuda')
t=time.time()
while time.time()-t<25:
    c=a@b; c=torch.relu(c); torch.cuda.synchronize()
