"""bf16 training steps of the HIP path against (a) the fp32 oracle and (b) the oracle in bf16-emulation mode (rounds
where the HIP path stores bf16): loss differences and per-tensor gradient rel-L2.  Run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import contextlib
import numpy as np, torch
from oracle import nets as O
from cfd_amd import nets, trainer, synthetic
import test_train_steps_gpu as T

def report(tag, mine, ref, prefix=""):
    rows = []
    for k, g in ref.items():
        if not k.startswith(prefix): continue
        name = k[len(prefix):]
        m = torch.from_numpy(mine[name]); gr = g.detach()
        if gr.abs().max() < 1e-6 * max(1.0, float(m.abs().max())) or (name.endswith(".bias") and (name[:-5] + ".bn.gamma") in mine):
            continue
        rows.append((name, T.rel_l2(m, gr)))
    worst = max(rows, key=lambda r: r[1])
    print(f"  {tag}: worst {worst[0]} {worst[1]:.3e} | " + ", ".join(f"{n}={e:.1e}" for n, e in rows if n.endswith("kernel")))

B = 2
x, u, _ = T.inputs(B)
for emu in (False, True):
    cm = O.bf16_emulation() if emu else contextlib.nullcontext()
    print("=== oracle", "bf16-emulated" if emu else "fp32")
    with cm:
        po = O.init_discriminator(seed=21, jitter=True)
        d = nets.Discriminator(dtype=torch.bfloat16); d.params.load_keras(O.to_numpy(po))
        tr = trainer.DiscriminatorTrainer(d)
        lab = np.ones((B, 4, 32, 1), np.float32)
        lref, gref = O.d_train_on_batch(po, O.KerasAdam(2e-3, 0.5), x, u, lab, return_grads=True)
        l = tr.train_on_batch(torch.from_numpy(x).cuda(), torch.from_numpy(u).cuda(), 1.0)
        print(f" D loss {l:.6f} vs {lref:.6f} rel {abs(l-lref)/max(1,abs(lref)):.2e}")
        report("D grads", T.grads_keras(d), gref)
        for arch in ("U", "P"):
            pg = O.init_generator(arch, seed=22, jitter=True); pd = O.init_discriminator(seed=23, jitter=True)
            g = nets.Generator(arch, dtype=torch.bfloat16); dd = nets.Discriminator(dtype=torch.bfloat16)
            g.params.load_keras(O.to_numpy(pg)); dd.params.load_keras(O.to_numpy(pd))
            gt = trainer.GanTrainer(g, dd)
            gen = torch.Generator().manual_seed(3)
            masks = [(torch.rand(s, generator=gen) < 0.5) for s in O.generator_dropout_shapes(arch, B)]
            ref, gr = O.gan_train_on_batch(pg, pd, O.KerasAdam(4e-4, 0.5), arch, x, lab, u, [m.float() for m in masks], return_grads=True)
            out = gt.train_on_batch(torch.from_numpy(x).cuda(), 1.0, torch.from_numpy(u).cuda(), masks=[m.to(torch.uint8).cuda() for m in masks])
            print(f" GAN {arch} losses {out} vs {ref} rel {[abs(a-b)/max(1,abs(b)) for a,b in zip(out,ref)]}")
            report(f"GAN {arch} G grads", T.grads_keras(g), gr, "g.")
        pc = O.init_cnn(seed=24, jitter=True)
        c = nets.CnnRegressor(dtype=torch.bfloat16); c.params.load_keras(O.to_numpy(pc))
        ct = trainer.CnnTrainer(c)
        x3, _, _ = T.inputs(3); y = np.array([[0.3], [0.9], [0.1]], np.float32)
        mask = torch.rand((3, 256), generator=torch.Generator().manual_seed(1)) < 0.5
        lref, gref = O.cnn_train_on_batch(pc, O.KerasAdam(1e-3, 0.5), x3, y, mask.float(), return_grads=True)
        l = ct.train_on_batch(torch.from_numpy(x3).cuda(), torch.from_numpy(y).cuda(), drop_mask=mask.to(torch.uint8).cuda())
        print(f" CNN loss {l:.6f} vs {lref:.6f} rel {abs(l-lref)/max(1,abs(lref)):.2e}")
        report("CNN grads", T.grads_keras(c), gref)
