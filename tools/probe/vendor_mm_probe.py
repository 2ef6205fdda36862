import torch
a = torch.randn(8192, 8192, device="cuda").to(torch.bfloat16); b = torch.randn(8192, 8192, device="cuda").to(torch.bfloat16)
for _ in range(3): c = a @ b.t()
x = torch.randn(32768, 3072, device="cuda").to(torch.bfloat16); w = torch.randn(768, 3072, device="cuda").to(torch.bfloat16)
for _ in range(3): y = x @ w.t()
torch.cuda.synchronize()
