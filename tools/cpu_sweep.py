import sys, time, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from oracle import svt_oracle as O
from svt_speechbrain_amd import weights as W
from svt_speechbrain_amd.config import PRESETS
cfg = PRESETS["wav2vec2-base"]; sd = W.seeded_encoder_state_dict(cfg); hd = W.seeded_head_state_dict(768)
g = torch.Generator().manual_seed(1986); wav = (0.1*torch.randn(4,160000,generator=g)).clamp_(-1,1)
print("cpu_count", os.cpu_count(), flush=True)
os.system("lscpu | grep -E 'Model name|Socket|Core|Thread' ")
for nt in [16, 32, 64, 128]:
    torch.set_num_threads(nt)
    def one():
        with torch.no_grad():
            f = O.encoder_forward(sd, cfg, wav); lg = O.head_forward(f, hd["w.weight"], hd["w.bias"])
    one(); t=time.perf_counter(); one(); dt=time.perf_counter()-t
    print(nt, "threads:", round(4/dt,3), "clips/s", flush=True)
