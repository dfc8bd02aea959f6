import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import torch.multiprocessing as mp


def worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as td
    from neuralplda_amd import dist as ndist, models, train
    torch.cuda.set_device(0)
    ndist.init("gloo")

    class Conf:
        xvector_dim, layer1_LDA_dim, layer2_PLDA_spkfactor_dim = 512, 150, 150
        beta, alpha, device, loss, log_interval = [99.0, 199.0], 15.0, "cuda", "SoftCdet", 1
    torch.manual_seed(0)
    m = models.NeuralPlda(Conf()).cuda()
    ndist.make_data_parallel(m)
    step = train.FusedTrainStep(m, 1e-3, weight_decay=1e-5, batch_size=64, graph=False)
    gen = torch.Generator(device="cuda").manual_seed(5)
    X1 = torch.randn(64, 512, device="cuda", generator=gen); X2 = torch.randn(64, 512, device="cuda", generator=gen)
    T = (torch.rand(64, device="cuda", generator=gen) < 0.3).float()
    nt = T.sum().double()
    gc = torch.stack([nt, 64.0 - nt])

    def cmp(tag, t):
        t = t.detach().cpu()
        o = [None] * world
        td.all_gather_object(o, t)
        if rank == 0:
            d = (o[0].double() - o[1].double()).abs().max().item()
            print(tag, "max diff between ranks", d, flush=True)
    for it in range(3):
        lo, hi = ndist.shard_bounds(64, world, rank)
        if it == 2:
            lo, hi = (0, 1) if rank == 0 else (1, 1)
            nt2 = T[:1].sum().double(); gcc = torch.stack([nt2, 1.0 - nt2])
        else:
            gcc = gc
        step(X1[lo:hi].contiguous(), X2[lo:hi].contiguous(), T[lo:hi].contiguous(), global_counts=gcc)
        torch.cuda.synchronize()
        cmp(f"it{it} flat", step._flat)
        cmp(f"it{it} step", step.step_count)
        cmp(f"it{it} m", step.m)
        cmp(f"it{it} v", step.v)
        for k, v in m.state_dict().items():
            cmp(f"it{it} {k}", v)
    td.destroy_process_group()


if __name__ == "__main__":
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(worker, args=(2, port), nprocs=2, join=True)
