"""Experiment: the B = 16 headline batch as TWO concurrent half-batches (two engine handles, two HIP streams, two host threads) - do the
launches of one half fill the partly empty rounds of the other's?   python tools/two_streams.py [steps] [T]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from ucdir_amd import networks, model as umodel
from ucdir_amd.weights import synth_inputs, synth_state_dict

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
T = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device("cuda", 0)


def make():
    net = networks.define_G(bench.sid_opt())
    sd = synth_state_dict(net.denoise_fn.cfg, 0)
    umodel.load_checkpoint_state(net, {k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    net = net.to(dev).eval()
    net.set_new_noise_schedule(dict(schedule="linear", n_timestep=T, linear_start=1e-6, linear_end=0.4), dev)
    return net


def run(nets, conds, streams, n):
    def work(i):
        with torch.cuda.stream(streams[i]), torch.no_grad():
            for _ in range(n):
                nets[i].super_resolution(conds[i], False)
        streams[i].synchronize()
    th = [threading.Thread(target=work, args=(i,)) for i in range(len(nets))]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


for parts in (1, 2, 4):
    B = 16 // parts
    nets = [make() for _ in range(parts)]
    conds = [torch.from_numpy(synth_inputs(B, 256, 256, seed=i)[0]).to(dev) for i in range(parts)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(parts)]
    run(nets, conds, streams, 1)
    dt = run(nets, conds, streams, steps)
    print("%d concurrent restoration(s) of B = %2d: %.2f img/s" % (parts, B, 16 * steps / dt)); sys.stdout.flush()
    del nets
    torch.cuda.empty_cache()
