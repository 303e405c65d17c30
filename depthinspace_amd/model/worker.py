"""Host-side mirror of /root/reference/model/worker.py: the experiment loop around the hot path.

Kept: class name, constructor signature (+ two optional keyword extensions), the overridable hooks
`get_train_set / get_test_sets / net_forward / loss_forward / callback_*`, `copy_data`,
`read_optical_flow`, `train_epoch`, `test_epoch`, `train`, the checkpoint file layout
(`state.dict`, `net_%04d.params`), `metrics.json`.
Changed on purpose: no per-phase torch.cuda.synchronize() in the step (reference :510,523,529,538),
loss terms are accumulated on the device and read back once per logging interval (reference reads
`err.item()` for every term every step, :551), LCN is applied only to the keys the hot path reads
(`im0`; the reference also normalises im1..3, ambient0 and the pattern copy that nothing consumes, :434-452).
Settings come from a `synth.Settings`-like object instead of config.json + settings.pkl when given.
New (the reference is single-GPU, model/worker.py:131,178,449): data parallelism.  Started under torchrun
(trainer.init_distributed), every rank runs this loop on its own shard of the tracks (seed + rank), gradients are
all-reduced by the optimiser (trainer.FlatAdam: bucketed, overlapped with backward), rank 0 alone logs and writes
checkpoints / metrics.json, and the test metrics are merged over the ranks.
"""
import json
import logging
import os
import pickle
import random
import time
from pathlib import Path

import numpy as np
import torch

from . import networks
from .multi_frame_networks import FlowDict


class ShardSampler(torch.utils.data.Sampler):
    """indices of rank r of `world`: a permutation common to all ranks (seeded by seed + epoch) dealt round-robin, every
    rank the same count (the surplus is dropped so no rank runs a collective the others do not)."""

    def __init__(self, n, rank, world, shuffle, seed=0, epoch=0):
        self.n, self.rank, self.world, self.shuffle, self.seed, self.epoch = n, rank, world, shuffle, seed, epoch

    def __iter__(self):
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            idx = torch.randperm(self.n, generator=g).tolist()
        else:
            idx = list(range(self.n))
        per = self.n // self.world if self.shuffle else -(-self.n // self.world)
        mine = idx[self.rank::self.world]
        return iter(mine[:per] if self.shuffle else mine)

    def __len__(self):
        if self.shuffle:
            return self.n // self.world
        return len(range(self.rank, self.n, self.world))


def split_paths(sample_paths, data_type):
    """reference model/worker.py:167-176: synthetic = first 512 valid, next 512 test, rest train; real = every 8th test.
    A synthetic directory with fewer than 1025 tracks (tests, demos) is split in the same 1:1:16 proportion."""
    sample_paths = list(sample_paths)
    n = len(sample_paths)
    if data_type == 'real':
        test = sample_paths[4::8]
        ts = set(test)
        return [p for p in sample_paths if p not in ts], test, []
    if n > 2 ** 10:
        return sample_paths[2 ** 10:], sample_paths[2 ** 9:2 ** 10], sample_paths[0:2 ** 9]
    k = max(1, n // 18)
    return sample_paths[2 * k:], sample_paths[k:2 * k], sample_paths[0:k]


class StopWatch(object):
    """reference :69-94"""

    def __init__(self):
        self.timings = {}
        self.starts = {}

    def start(self, name):
        self.starts[name] = time.time()

    def stop(self, name):
        self.timings.setdefault(name, []).append(time.time() - self.starts[name])

    def get(self, name=None, reduce=np.sum):
        if name is not None:
            return reduce(self.timings[name])
        return {k: reduce(v) for k, v in self.timings.items()}

    def __repr__(self):
        return ', '.join(['%s: %f[s]' % (k, v) for k, v in self.get().items()])


class Worker(object):
    def __init__(self, args, seed=42, test_batch_size=4, num_workers=4, save_frequency=1, train_device=None,
                 test_device=None, max_train_iter=-1, settings=None, output_dir=None, data_root=None, use_graph=None):
        self.use_pseudo_gt = args.use_pseudo_gt
        self.lcn_radius = args.lcn_radius
        self.track_length = args.track_length
        self.data_type = args.data_type
        self.architecture = args.architecture
        self.epochs = args.epochs
        self.warmup_epochs = args.warmup_epochs
        self.seed = seed
        self.train_batch_size = args.train_batch_size
        self.test_batch_size = test_batch_size
        self.num_workers = num_workers
        self.save_frequency = save_frequency
        # data parallelism: one process per GPU (torchrun); the reference hard-codes 'cuda:0'
        dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()
        self.rank = torch.distributed.get_rank() if dist_on else 0
        self.world_size = torch.distributed.get_world_size() if dist_on else 1
        self.is_main = self.rank == 0
        default_dev = f'cuda:{torch.cuda.current_device()}' if torch.cuda.is_available() else 'cuda:0'
        self.train_device = train_device or default_dev
        self.test_device = test_device or default_dev
        self.max_train_iter = max_train_iter
        self.current_epoch = 0
        # capture the step in hipGraphs (trainer.GraphedStep) instead of ~470 eager launches per step
        self.use_graph = (os.environ.get('DIS_TRAIN_GRAPH', '0') == '1') if use_graph is None else bool(use_graph)
        self.allow_optimizer_reset = os.environ.get('DIS_ALLOW_OPTIMIZER_RESET', '0') == '1'
        # training-time augmentation (reference data/dataset.py:128-186, data_aug=True for the train set) runs on the
        # device after the host->device copy; on for tracks read from DATA_DIR, off for the in-memory synthetic scenes
        self.device_aug = data_root is not None or (settings is None)
        if os.environ.get('DIS_DEVICE_AUG') is not None:
            self.device_aug = os.environ['DIS_DEVICE_AUG'] == '1'
        self._aug_rng = np.random.RandomState((seed + 1234 + self.rank) % (2 ** 31))
        self.data_root = Path(data_root) if data_root is not None else None

        if settings is None:
            # reference :152-166: config.json -> DATA_DIR/settings.pkl (here also settings.npz, the h5py-free mirror of
            # the on-disk schema, data/dataset.py)
            if self.data_root is None:
                config_path = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..', 'config.json'))
                with open(config_path) as fp:
                    config = json.load(fp)
                self.data_root = Path(config['DATA_DIR'])
                output_dir = output_dir or config['OUTPUT_DIR']
            if (self.data_root / 'settings.npz').exists():
                from ..data.dataset import load_settings
                settings = load_settings(str(self.data_root))
            else:
                with open(str(self.data_root / 'settings.pkl'), 'rb') as f:
                    s = pickle.load(f)
                from ..synth import Settings
                settings = Settings(s['imsize'], s['K'], s['baseline'], s['pattern'])
        if self.data_root is not None:
            self.settings_path = str(self.data_root)
            sample_paths = sorted(str(p) for p in self.data_root.glob('0*/'))
            self.train_paths, self.test_paths, self.valid_paths = split_paths(sample_paths, self.data_type)
        self.settings = settings
        self.baseline = settings.baseline
        self.K = np.asarray(settings.K, dtype=np.float32)
        self.Ki = np.linalg.inv(self.K)
        self.imsizes = [tuple(settings.imsize)]
        for _ in range(3):
            self.imsizes.append((int(self.imsizes[-1][0] / 2), int(self.imsizes[-1][1] / 2)))
        self.ref_pattern = settings.pattern
        self.output_dir = Path(output_dir) if output_dir is not None else None
        self.lcn_in = networks.LCN(self.lcn_radius, 0.05)
        self.metric_data = {}
        self.init_seed()
        if self.output_dir is not None:
            self.setup_experiment()

    # ------------------------------------------------------------------ experiment plumbing
    def setup_experiment(self):
        self.exp_output_dir = self.output_dir / self.architecture
        if self.is_main:
            self.exp_output_dir.mkdir(parents=True, exist_ok=True)
            if logging.root:
                del logging.root.handlers[:]
            logging.basicConfig(level=logging.INFO,
                                handlers=[logging.FileHandler(str(self.exp_output_dir / 'train.log')),
                                          logging.StreamHandler()],
                                format='%(relativeCreated)d:%(levelname)s:%(process)d-%(processName)s: %(message)s')
        else:
            logging.basicConfig(level=logging.WARNING)
        self.metric_path = self.exp_output_dir / 'metrics.json'
        if self.is_main and self.metric_path.exists():
            with open(str(self.metric_path), 'r') as fp:
                self.metric_data = json.load(fp)

    def init_seed(self, seed=None):
        if seed is not None:
            self.seed = seed
        np.random.seed(self.seed)
        random.seed(self.seed)
        torch.manual_seed(self.seed)

    def metric_add_train(self, epoch, key, val):
        self.metric_data.setdefault(str(epoch), {}).setdefault('train', {})[str(key)] = val

    def metric_add_test(self, epoch, set_idx, key, val):
        self.metric_data.setdefault(str(epoch), {}).setdefault('test', {}).setdefault(str(set_idx), {})[str(key)] = val

    def metric_save(self):
        if self.output_dir is not None and self.is_main:
            with open(str(self.metric_path), 'w') as fp:
                json.dump(self.metric_data, fp, indent=2)

    def get_net_path(self, epoch, root=None):
        root = self.exp_output_dir if root is None else root
        return root / f'net_{epoch:04d}.params'

    def format_err_str(self, errs, div=1):
        err = sum(errs)
        if len(errs) > 1:
            return f'{err / div:0.4f}=' + '+'.join([f'{e / div:0.4f}' for e in errs])
        return f'{err / div:0.4f}'

    # ------------------------------------------------------------------ hooks
    def get_train_set(self):
        raise NotImplementedError()

    def get_test_sets(self):
        raise NotImplementedError()

    def net_forward(self, net, train):
        raise NotImplementedError()

    def loss_forward(self, output, train, flow_out):
        raise NotImplementedError()

    def callback_train_post_backward(self, net, errs, output, epoch, batch_idx, masks):
        pass

    def callback_train_start(self, epoch):
        pass

    def callback_train_stop(self, epoch, loss):
        pass

    def callback_train_new_epoch(self, epoch, net, optimizer):
        pass

    def callback_test_start(self, epoch, set_idx):
        pass

    def callback_test_add(self, epoch, set_idx, batch_idx, n_batches, output, masks):
        pass

    def callback_test_stop(self, epoch, set_idx, loss):
        pass

    # ------------------------------------------------------------------ the hot path
    def copy_data(self, data, device, requires_grad, train):
        """reference :418-452.  (bs,tl,...) loader tensors -> self.data with (tl,bs,...) device tensors;
        `im0` becomes cat(LCN(im0), im0) and `std0` is stored."""
        self.data = {}
        for key, val in data.items():
            val = torch.as_tensor(val)
            if len(val.shape) > 2:
                val = val.transpose(0, 1)
            self.data[key] = val.to(device).contiguous()
        if '_flow_stacked' in self.data:
            # all ordered-pair flows delivered as one (bs, tl*tl, 2, H, W) tensor (entry i*tl+j): after the
            # transpose it is exactly the stacked layout the network consumes; flow_ij become views of it
            fs = self.data['_flow_stacked']
            tl = int(round(fs.shape[0] ** 0.5))
            for i in range(tl):
                for j in range(tl):
                    if i != j:
                        self.data[f'flow_{i}{j}'] = fs[i * tl + j].unsqueeze(0)
        if train and self.device_aug:
            self._augment_on_device(device)
        im = self.data['im0']
        tl, bs = im.shape[0], im.shape[1]
        im_lcn, im_std = self.lcn_in(im.view(-1, *im.shape[2:]))
        self.data['std0'] = im_std.view(tl, bs, *im.shape[2:])
        self.data['im0'] = torch.cat((im_lcn.view(tl, bs, *im.shape[2:]), im), dim=2)  # memory op

    def draw_aug(self, n_images):
        """host-side random choices of one step's augmentation (reference data/data_manipulation.py:161-177): CPU tensors"""
        from .. import ops
        params = torch.from_numpy(ops.draw_augment_params(n_images, self._aug_rng))
        seed = torch.tensor([int(self._aug_rng.randint(0, 2 ** 31 - 1)) * 2654435761 + self.rank], dtype=torch.int64)
        return params, seed

    def _augment_on_device(self, device):
        """blur / noise / salt-and-pepper of im0 and ambient0 (reference data/dataset.py:128-186 does it with cv2 in the
        loader processes); a captured step supplies the draws as `_aug_params` / `_aug_seed`"""
        from .. import ops
        im, amb = self.data['im0'], self.data['ambient0']
        n = im.numel() // (im.shape[-1] * im.shape[-2])
        params, seed = self.data.get('_aug_params'), self.data.get('_aug_seed')
        if params is None:
            params, seed = self.draw_aug(n)
            params, seed = params.to(device), seed.to(device)
        self.data['im0'], self.data['ambient0'] = ops.augment(im, amb, params, seed)

    def read_optical_flow(self, train):
        """reference :457-465"""
        tl = self.data['ambient0'].shape[0]
        out = FlowDict()
        for i in range(tl):
            for j in range(tl):
                if i != j:
                    out[f'flow_{i}{j}'] = self.data[f'flow_{i}{j}'][0]
        out.stacked = self.data.get('_flow_stacked', None)
        return out

    def train_step(self, net, optimizer, data):
        """One iteration of the reference loop body (:499-539): copy_data, zero_grad, flow dict, net_forward,
        loss_forward, backward, optimizer.step.  Returns the list of loss terms (device scalars)."""
        self.copy_data(data, device=self.train_device, requires_grad=False, train=True)
        optimizer.zero_grad()
        flow_output = self.read_optical_flow(train=True)
        output = self.net_forward(net, flow_output)
        errs = self.loss_forward(output, True, flow_output)
        if isinstance(errs, dict):
            errs = errs['errs']
        if not isinstance(errs, (list, tuple)):
            errs = [errs]
        tot = getattr(errs, 'total', None)   # (ops.LossTerms: the sum as one autograd node)
        (tot if tot is not None else sum(errs)).backward()
        optimizer.step()
        return errs, output

    def _loader(self, dset, batch_size, train, epoch):
        from ..data.dataset import collate
        sampler = ShardSampler(len(dset), self.rank, self.world_size, shuffle=train, seed=self.seed, epoch=epoch)
        return torch.utils.data.DataLoader(dset, batch_size=batch_size, sampler=sampler, num_workers=self.num_workers,
                                           drop_last=train, pin_memory=train, collate_fn=collate,
                                           worker_init_fn=self._seed_loader_worker)

    def _seed_loader_worker(self, worker_id):
        # per-rank, per-worker host RNG streams (frame permutation, augmentation): seed + rank as SURVEY section 8(e)
        np.random.seed((self.seed + 1000 * self.rank + worker_id + 7919 * self.current_epoch) % (2 ** 31))

    def train_epoch(self, epoch, net, optimizer, dset):
        """reference :479-566.  Every rank iterates its own shard; the optimiser all-reduces the gradients."""
        self.callback_train_start(epoch)
        stopwatch = StopWatch()
        logging.info('=' * 80)
        logging.info('Train epoch %d' % epoch)
        dset.current_epoch = epoch
        train_loader = self._loader(dset, self.train_batch_size, True, epoch)
        net = net.to(self.train_device)
        net.train()
        mean_loss = None
        n_done = 0
        stopwatch.start('total')
        for batch_idx, data in enumerate(train_loader):
            if self.max_train_iter > 0 and batch_idx > self.max_train_iter:
                break
            if self.use_graph:
                graphed = self._graphed_step(net, optimizer, data)
                data = dict(data)
                if self.device_aug:
                    im = data['im0']
                    data['_aug_params'], data['_aug_seed'] = self.draw_aug(im.shape[0] * im.shape[1])
                if self.data_type == 'real' and self.current_epoch < self.warmup_epochs:
                    for k in range(self.n_sgm_draws):  # host generator, as the reference
                        data[f'_sgm_noise{k}'] = 1.5 * torch.randn(data['sgm_disp'].shape)
                graphed.run(data)
                stacked = graphed.loss_buf[:graphed.nterms].clone()
                errs, output = list(stacked), None
            else:
                errs, output = self.train_step(net, optimizer, data)
                stacked = torch.stack([e.detach() for e in errs])
            self.callback_train_post_backward(net, errs, output, epoch, batch_idx, [])
            mean_loss = stacked if mean_loss is None else mean_loss + stacked
            n_done += 1
            if (epoch <= 1 and batch_idx < 128) or batch_idx % 16 == 0:
                logging.info(f'train e{epoch}: {batch_idx + 1}/{len(train_loader)}: '
                             f'loss={self.format_err_str([float(e) for e in stacked.cpu()])}')
        if 'cuda' in str(self.train_device):
            torch.cuda.synchronize()
        stopwatch.stop('total')
        logging.info('timings: %s' % stopwatch)
        self.last_epoch_stats = {'steps': n_done, 'seconds': float(stopwatch.get('total')),
                                 'step_mode': (self._graphed[1].mode if self.use_graph and getattr(self, '_graphed', None)
                                               else 'eager'),
                                 'frames_per_s': (n_done * self.train_batch_size * self.track_length * self.world_size /
                                                  max(float(stopwatch.get('total')), 1e-9))}
        if mean_loss is not None and self.world_size > 1:
            torch.distributed.all_reduce(mean_loss)
            mean_loss = mean_loss / self.world_size
        mean_loss = [float(v) / max(n_done, 1) for v in mean_loss.cpu()] if mean_loss is not None else []
        self.callback_train_stop(epoch, mean_loss)
        self.metric_add_train(epoch, 'loss', mean_loss)
        self.metric_save()
        logging.info(f'avg train_loss={self.format_err_str(mean_loss) if mean_loss else "n/a"}')
        return mean_loss

    def _graphed_step(self, net, optimizer, example_batch):
        """the captured step of this (net, optimizer, batch shapes): built once and kept across epochs (static buffers,
        warm-up steps and the capture are paid once; GraphedStep.key() re-captures when the set of loss terms changes)"""
        from ..trainer import GraphedStep
        sig = (id(net), id(optimizer), tuple((k, tuple(torch.as_tensor(v).shape)) for k, v in sorted(example_batch.items())))
        cached = getattr(self, '_graphed', None)
        if cached is not None and cached[0] == sig:
            return cached[1]
        gs = self._build_graphed_step(net, optimizer, example_batch)
        self._graphed = (sig, gs)
        return gs

    def _build_graphed_step(self, net, optimizer, example_batch):
        from ..trainer import GraphedStep
        ex = dict(example_batch)
        if self.device_aug:
            im = ex['im0']
            ex['_aug_params'], ex['_aug_seed'] = self.draw_aug(im.shape[0] * im.shape[1])
        if self.data_type == 'real':
            for k in range(self.n_sgm_draws):
                ex[f'_sgm_noise{k}'] = torch.zeros_like(torch.as_tensor(ex['sgm_disp']))
        # DIS_TRAIN_GRAPH=1 asked for the captured step explicitly: a failing capture raises instead of training eagerly
        return GraphedStep(self, net, optimizer, ex, use_graph=True,
                           strict=os.environ.get('DIS_TRAIN_GRAPH', '') == '1')

    def test(self, epoch, net, test_sets):
        errs = {}
        for test_set_idx, test_set in enumerate(test_sets):
            if (epoch + 1) % test_set.test_frequency == 0:
                errs[test_set.name] = self.test_epoch(epoch, test_set_idx, net, test_set.dset)
        return errs

    def test_epoch(self, epoch, set_idx, net, dset):
        """reference :587-653 (same forward/loss under no_grad); the ranks evaluate disjoint shards of the set, the loss
        means and the metrics of callback_test_* are merged over the ranks."""
        loader = self._loader(dset, self.test_batch_size, False, epoch)
        net = net.to(self.test_device)
        net.eval()
        sum_loss = None
        n = 0
        with torch.no_grad():
            self.callback_test_start(epoch, set_idx)
            for batch_idx, data in enumerate(loader):
                self.copy_data(data, device=self.test_device, requires_grad=False, train=False)
                flow_output = self.read_optical_flow(train=False)
                output = self.net_forward(net, flow_output)
                errs = self.loss_forward(output, False, flow_output)
                stacked = torch.stack([e.detach() for e in errs])
                sum_loss = stacked if sum_loss is None else sum_loss + stacked
                n += 1
                self.callback_test_add(epoch, set_idx, batch_idx, len(loader), output, [])
            cnt = torch.tensor([float(n)], device=self.test_device)
            if self.world_size > 1:
                if sum_loss is None:  # a rank whose shard is empty still takes part in the collectives
                    nt = torch.zeros(1, device=self.test_device)
                    torch.distributed.all_reduce(nt, op=torch.distributed.ReduceOp.MAX)
                    sum_loss = torch.zeros(int(nt), device=self.test_device)
                else:
                    nt = torch.tensor([float(sum_loss.numel())], device=self.test_device)
                    torch.distributed.all_reduce(nt, op=torch.distributed.ReduceOp.MAX)
                torch.distributed.all_reduce(sum_loss)
                torch.distributed.all_reduce(cnt)
            mean_loss = [float(v) / max(float(cnt), 1.0) for v in sum_loss.cpu()]
            self.callback_test_stop(epoch, set_idx, mean_loss)
        self.metric_add_test(epoch, set_idx, 'loss', mean_loss)
        self.metric_save()
        logging.info(f'test epoch {epoch}: avg test_loss={self.format_err_str(mean_loss)}')
        return mean_loss

    def train(self, net, optimizer, resume=False, scheduler=None):
        """reference :328-410 incl. the state.dict / net_%04d.params checkpoint layout.  optimizer.state_dict() is in
        torch.optim.Adam's own layout (trainer.FlatAdam), so state.dict files written by the reference resume here."""
        train_set = self.get_train_set()
        test_sets = self.get_test_sets()
        net = net.to(self.train_device)
        epoch = 0
        min_err = {ts.name: 1e9 for ts in test_sets}
        state_path = self.exp_output_dir / 'state.dict'
        if resume and state_path.exists():
            logging.info(f'Loading state from {state_path}')
            state = torch.load(str(state_path), map_location='cpu', weights_only=False)
            epoch = state['epoch'] + 1
            min_err = state.get('min_err', min_err)
            curr_state = net.state_dict()
            curr_state.update(state['state_dict'])
            net.load_state_dict(curr_state)
            try:
                optimizer.load_state_dict(state['optimizer'])
            except Exception as e:
                # the reference only logs a warning here (:356-360); silently restarting Adam's moments and bias correction
                # in the middle of a run is a training bug, so it has to be asked for
                if not self.allow_optimizer_reset:
                    raise RuntimeError(f'cannot restore the optimizer from {state_path} ({type(e).__name__}: {e}); '
                                       f'set DIS_ALLOW_OPTIMIZER_RESET=1 to resume with a fresh optimizer') from e
                logging.error(f'cannot load optimizer from state_dict ({e}): resuming with a FRESH optimizer')
            if 'cpu_rng_state' in state:
                torch.set_rng_state(state['cpu_rng_state'])
            if 'gpu_rng_state' in state and torch.cuda.is_available():
                torch.cuda.set_rng_state(state['gpu_rng_state'].cpu())
        if self.world_size > 1:
            if hasattr(optimizer, 'broadcast_parameters'):
                optimizer.broadcast_parameters(0)  # identical replicas whatever each rank initialised / loaded
            # ... and ONE resume point: without a shared file system only rank 0 may have found state.dict; the epoch decides
            # the number of epochs left and the set of loss terms (warm-up terms), i.e. the collectives every rank issues
            meta = [epoch, min_err]
            torch.distributed.broadcast_object_list(meta, src=0)
            epoch, min_err = meta
        for epoch in range(epoch, self.epochs):
            self.current_epoch = epoch
            self.callback_train_new_epoch(epoch, net, optimizer)
            self.train_epoch(epoch, net, optimizer, train_set)
            errs = self.test(epoch, net, test_sets)
            if (epoch + 1) % self.save_frequency == 0 and self.is_main:
                state_dict = {'epoch': epoch, 'min_err': min_err, 'state_dict': net.state_dict(),
                              'optimizer': optimizer.state_dict(), 'cpu_rng_state': torch.get_rng_state(),
                              'gpu_rng_state': torch.cuda.get_rng_state()}
                logging.info(f'save state to {state_path}')
                _atomic_save(state_dict, state_path)
                for name in errs:
                    err = sum(errs[name])
                    if err < min_err[name]:
                        min_err[name] = err
                        _atomic_save(state_dict, self.exp_output_dir / f'state_set_{name}_best.dict')
                _atomic_save(net.state_dict(), self.get_net_path(epoch))
            if scheduler is not None:
                scheduler.step()
            if self.world_size > 1:
                torch.distributed.barrier()

    def retest(self, net, epoch=-1):
        """reference :289-302"""
        epochs = range(self.epochs) if epoch < 0 else [epoch]
        test_sets = self.get_test_sets()
        for epoch in epochs:
            net_path = self.get_net_path(epoch)
            if net_path.exists():
                net.load_state_dict(torch.load(str(net_path), map_location='cpu'))
                self.test(epoch, net, test_sets)

    def do(self, net, optimizer, cmd='resume', epoch=-1, scheduler=None):
        """reference :267-287"""
        if cmd == 'retrain':
            self.train(net, optimizer, resume=False, scheduler=scheduler)
        elif cmd == 'resume':
            self.train(net, optimizer, resume=True, scheduler=scheduler)
        elif cmd == 'retest':
            self.retest(net, epoch=epoch)
        elif cmd == 'test_init':
            self.test(-1, net, self.get_test_sets())
        else:
            raise Exception('invalid cmd')


def _atomic_save(obj, path):
    """torch.save to a temporary file in the same directory, then rename over `path`: a crash while writing leaves the
    previous checkpoint intact (the reference overwrites its only state.dict in place, model/worker.py:376-402)"""
    tmp = str(path) + '.tmp'
    torch.save(obj, tmp)
    os.replace(tmp, str(path))


class TestSet(object):
    """reference data/base_dataset.py:29-37"""

    def __init__(self, name, dset, test_frequency=1):
        self.name = name
        self.dset = dset
        self.test_frequency = test_frequency


class TestSets(list):
    def append(self, name, dset, test_frequency=1):
        super().append(TestSet(name, dset, test_frequency))
