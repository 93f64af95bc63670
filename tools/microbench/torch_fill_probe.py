import torch
buf = torch.empty(64 * 512 * 512 * 1125 // 4, dtype=torch.float32, device="cuda")
for _ in range(5):
    buf.fill_(1.0)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    buf.fill_(1.0)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"torch fill {ms:.3f} ms {buf.numel()*4/ms/1e9:.2f} TB/s")
