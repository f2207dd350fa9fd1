"""Debug helper: run one 192^3 (or --d) training step twice and list the gradient tensors that differ bit-wise."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_unet_gpu import make_net, make_vdm, DEV, vdm_model_mod
from vdm4cdm_amd.data import SyntheticAstroDataModule
import vdm4cdm_amd.unet_hip as uh
D = int(sys.argv[1]) if len(sys.argv) > 1 else 192
net = make_net(D=D, chs=(32, 64, 128, 256), precision="bf16", dropout=0.1, seed=3)
vdm = make_vdm(net).to(DEV).train()
b = SyntheticAstroDataModule(cropsize=D, batch_size=2, seed=1000)._make_batch(1000, 2)
batch = {"x": b["x"].to(DEV), "conditioning": b["conditioning"].to(DEV), "conditioning_values": [b["conditioning_values"][0].to(DEV)]}
outs = []
for rep in range(6):
    torch.manual_seed(11)
    uh._seed_counter[0] = 0
    vdm_model_mod.reset_train_generators()
    vdm.zero_grad()
    loss = vdm.training_step(batch, 0)
    loss.backward()
    torch.cuda.synchronize()
    outs.append((loss.detach().clone(), net.flat.grad.detach().clone()))
for k in (1, 2, 3, 4, 5):
    diff = [(name, (net.view(name, outs[0][1]) - net.view(name, outs[k][1])).abs().max().item(), net.view(name, outs[0][1]).abs().max().item())
            for name in net.spec.items if not torch.equal(net.view(name, outs[0][1]), net.view(name, outs[k][1]))]
    print(f"rep {k}: loss equal {torch.equal(outs[0][0], outs[k][0])}; {len(diff)} tensors differ")
    for d in diff[:3] + diff[-2:]:
        print("   %-40s max|d| %.3e of %.3e" % d)
