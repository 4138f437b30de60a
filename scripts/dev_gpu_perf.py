"""Developer timing on a GPU box: batch encode/decode stage times (not the bench; see bench.py)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import alice_codec_amd as a

W,H,F = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (1920,1080,64)))
B = int(sys.argv[4]) if len(sys.argv) > 4 else 1
k = int(sys.argv[5]) if len(sys.argv) > 5 else 1
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1234)
t = torch.arange(F, device=dev).view(F,1,1,1).float(); y = torch.arange(H, device=dev).view(1,H,1,1).float(); x = torch.arange(W, device=dev).view(1,1,W,1).float()
s = torch.tensor([23.,31.,17.], device=dev).view(1,1,1,3); ph = torch.tensor([0.,1.,2.], device=dev).view(1,1,1,3)
chunks = []
for b in range(B):
    base = 128 + 90*torch.sin((x+2*t+7*b)/s+ph)*torch.cos((y-t)/(0.7*s))
    noise = torch.randint(-4,5,(F,H,W,3),device=dev,generator=g)
    chunks.append((base+noise).clamp(0,255).to(torch.uint8))
rgb = torch.stack(chunks).contiguous()
out = torch.empty_like(rgb)
torch.cuda.synchronize()
bt = a.Batch(W,H,F,B,80,a.WaveletType(k))
st = torch.cuda.current_stream().cuda_stream
for it in range(2):
    t0=time.time(); bt.encode(rgb.data_ptr(), st); sizes = bt.encode_finish(); t1=time.time()
    bt.decode(bt.alc_ptr(0), bt.alc_stride, out.data_ptr(), st); bt.decode_finish(); t2=time.time()
    print(f"iter {it}: enc {t1-t0:.3f}s dec {t2-t1:.3f}s sizes {sizes[:2]} bytes/px {float(sizes.sum())/(B*W*H*F):.3f}", bt.stage_ms(), flush=True)
mpix = B*W*H*F/1e6
print(f"B={B} {W}x{H}x{F}: enc {mpix/(t1-t0):.1f} Mpix/s dec {mpix/(t2-t1):.1f} Mpix/s  combined {2*mpix/(t2-t0):.1f}")
d = (out.int()-rgb.int()).float(); print("psnr-ish mse", float((d*d).mean()))
