"""2 ranks on one GPU (gloo), full-size DIS-MF eager overlapped step: which gradients / parameters go non-finite, per step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import torch.multiprocessing as mp


def rank_main(rank, world, port):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import bench
    from depthinspace_amd import synth
    from depthinspace_amd.model import multi_frame_networks, multi_frame_worker
    from depthinspace_amd.trainer import FlatAdam, GraphedStep, init_distributed
    init_distributed('gloo')
    dev = torch.device('cuda:0')
    H, W, TL = 512, 432, 4
    settings = synth.make_settings(H, W)
    torch.manual_seed(0)
    net = multi_frame_networks.FuseNet(imsize=(H, W), K=settings.K, baseline=settings.baseline, track_length=TL, max_disp=128).to(dev)
    worker = multi_frame_worker.Worker(bench.make_args(4), settings=settings, train_device=str(dev))
    worker.build_losses(device=dev)
    worker.current_epoch = 2
    opt = FlatAdam(net.parameters(), lr=1e-4, world_size=world)
    if os.environ.get('NO_OVERLAP'):
        opt.overlap = False
    batch = bench.make_device_batch(settings, 4, 1234 + rank, dev)
    stepper = GraphedStep(worker, net, opt, batch, use_graph=False, warmup=1)
    names = [n for n, _ in net.named_parameters()]
    for step in range(4):
        stepper.run()
        torch.cuda.synchronize()
        bad_g = [names[i] for i, (p, off) in enumerate(zip(opt.params, opt.offsets))
                 if not bool(torch.isfinite(opt.flat_g[off:off + p.numel()]).all())]
        bad_p = [names[i] for i, p in enumerate(opt.params) if not bool(torch.isfinite(p).all())]
        print(f'rank {rank} step {step}: losses {[round(float(l), 5) for l in stepper.losses()]} non-finite grads {bad_g[:6]} '
              f'({len(bad_g)}) params ({len(bad_p)}) {bad_p[:4]}', flush=True)
    torch.distributed.barrier()


if __name__ == '__main__':
    mp.set_start_method('spawn')
    ps = [mp.Process(target=rank_main, args=(r, 2, 29611)) for r in range(2)]
    [p.start() for p in ps]
    [p.join() for p in ps]
