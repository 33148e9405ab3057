import torch
def t(fn, it=10):
    fn(); torch.cuda.synchronize()
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/it
x=torch.randn(256,64,224,224,device='cuda'); y=torch.empty_like(x)
gb=x.numel()*4/1e9
print('sum   read  %.2f TB/s'%(gb/t(lambda: x.sum())))
print('clone r+w   %.2f TB/s'%(2*gb/t(lambda: y.copy_(x))))
print('fill  write %.2f TB/s'%(gb/t(lambda: y.fill_(1.0))))
print('add   2r+w  %.2f TB/s'%(3*gb/t(lambda: torch.add(x,y,out=y))))
print('mean over (0,2,3) read %.2f TB/s'%(gb/t(lambda: x.mean(dim=(0,2,3)))))
