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
    def __init__(self, args, seed=42, test_batch_size=4, num_workers=4, save_frequency=1, train_device='cuda:0',
                 test_device='cuda:0', max_train_iter=-1, settings=None, output_dir=None):
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
        self.train_device = train_device
        self.test_device = test_device
        self.max_train_iter = max_train_iter
        self.current_epoch = 0

        if settings is None:
            # reference :152-166: config.json -> DATA_DIR/settings.pkl
            config_path = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..', 'config.json'))
            with open(config_path) as fp:
                config = json.load(fp)
            with open(str(Path(config['DATA_DIR']) / 'settings.pkl'), 'rb') as f:
                s = pickle.load(f)
            from ..synth import Settings
            settings = Settings(s['imsize'], s['K'], s['baseline'], s['pattern'])
            output_dir = output_dir or config['OUTPUT_DIR']
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
        self.exp_output_dir.mkdir(parents=True, exist_ok=True)
        self.metric_path = self.exp_output_dir / 'metrics.json'
        if self.metric_path.exists():
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
        if self.output_dir is not None:
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
        im = self.data['im0']
        tl, bs = im.shape[0], im.shape[1]
        im_lcn, im_std = self.lcn_in(im.view(-1, *im.shape[2:]))
        self.data['std0'] = im_std.view(tl, bs, *im.shape[2:])
        self.data['im0'] = torch.cat((im_lcn.view(tl, bs, *im.shape[2:]), im), dim=2)  # memory op

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
        sum(errs).backward()
        optimizer.step()
        return errs, output

    def train_epoch(self, epoch, net, optimizer, dset):
        """reference :479-566"""
        self.callback_train_start(epoch)
        stopwatch = StopWatch()
        logging.info('=' * 80)
        logging.info('Train epoch %d' % epoch)
        dset.current_epoch = epoch
        from ..data.dataset import collate
        train_loader = torch.utils.data.DataLoader(dset, batch_size=self.train_batch_size, shuffle=True,
                                                   num_workers=self.num_workers, drop_last=True, pin_memory=True,
                                                   collate_fn=collate)
        net = net.to(self.train_device)
        net.train()
        mean_loss = None
        n_done = 0
        stopwatch.start('total')
        for batch_idx, data in enumerate(train_loader):
            if self.max_train_iter > 0 and batch_idx > self.max_train_iter:
                break
            errs, output = self.train_step(net, optimizer, data)
            self.callback_train_post_backward(net, errs, output, epoch, batch_idx, [])
            stacked = torch.stack([e.detach() for e in errs])
            mean_loss = stacked if mean_loss is None else mean_loss + stacked
            n_done += 1
            if (epoch <= 1 and batch_idx < 128) or batch_idx % 16 == 0:
                logging.info(f'train e{epoch}: {batch_idx + 1}/{len(train_loader)}: '
                             f'loss={self.format_err_str([float(e) for e in stacked.cpu()])}')
        stopwatch.stop('total')
        logging.info('timings: %s' % stopwatch)
        mean_loss = [float(v) / max(n_done, 1) for v in mean_loss.cpu()] if mean_loss is not None else []
        self.callback_train_stop(epoch, mean_loss)
        self.metric_add_train(epoch, 'loss', mean_loss)
        self.metric_save()
        logging.info(f'avg train_loss={self.format_err_str(mean_loss) if mean_loss else "n/a"}')
        return mean_loss

    def test(self, epoch, net, test_sets):
        errs = {}
        for test_set_idx, test_set in enumerate(test_sets):
            if (epoch + 1) % test_set.test_frequency == 0:
                errs[test_set.name] = self.test_epoch(epoch, test_set_idx, net, test_set.dset)
        return errs

    def test_epoch(self, epoch, set_idx, net, dset):
        """reference :587-653 (same forward/loss under no_grad)"""
        from ..data.dataset import collate
        loader = torch.utils.data.DataLoader(dset, batch_size=self.test_batch_size, shuffle=False,
                                             num_workers=self.num_workers, drop_last=False, collate_fn=collate)
        net = net.to(self.test_device)
        net.eval()
        mean_loss = None
        n = 0
        with torch.no_grad():
            self.callback_test_start(epoch, set_idx)
            for batch_idx, data in enumerate(loader):
                self.copy_data(data, device=self.test_device, requires_grad=False, train=False)
                flow_output = self.read_optical_flow(train=False)
                output = self.net_forward(net, flow_output)
                errs = self.loss_forward(output, False, flow_output)
                stacked = torch.stack([e.detach() for e in errs])
                mean_loss = stacked if mean_loss is None else mean_loss + stacked
                n += 1
                self.callback_test_add(epoch, set_idx, batch_idx, len(loader), output, [])
        mean_loss = [float(v) / max(n, 1) for v in mean_loss.cpu()]
        self.callback_test_stop(epoch, set_idx, mean_loss)
        self.metric_add_test(epoch, set_idx, 'loss', mean_loss)
        self.metric_save()
        return mean_loss

    def train(self, net, optimizer, resume=False, scheduler=None):
        """reference :328-410 incl. the state.dict / net_%04d.params checkpoint layout."""
        train_set = self.get_train_set()
        test_sets = self.get_test_sets()
        net = net.to(self.train_device)
        epoch = 0
        min_err = {ts.name: 1e9 for ts in test_sets}
        state_path = self.exp_output_dir / 'state.dict'
        if resume and state_path.exists():
            state = torch.load(str(state_path), weights_only=False)
            epoch = state['epoch'] + 1
            min_err = state.get('min_err', min_err)
            curr_state = net.state_dict()
            curr_state.update(state['state_dict'])
            net.load_state_dict(curr_state)
            try:
                optimizer.load_state_dict(state['optimizer'])
            except Exception:
                logging.info('Warning: cannot load optimizer from state_dict')
            if 'cpu_rng_state' in state:
                torch.set_rng_state(state['cpu_rng_state'])
        for epoch in range(epoch, self.epochs):
            self.current_epoch = epoch
            self.callback_train_new_epoch(epoch, net, optimizer)
            self.train_epoch(epoch, net, optimizer, train_set)
            errs = self.test(epoch, net, test_sets)
            if (epoch + 1) % self.save_frequency == 0:
                state_dict = {'epoch': epoch, 'min_err': min_err, 'state_dict': net.state_dict(),
                              'optimizer': optimizer.state_dict(), 'cpu_rng_state': torch.get_rng_state()}
                torch.save(state_dict, str(state_path))
                for name in errs:
                    err = sum(errs[name])
                    if err < min_err[name]:
                        min_err[name] = err
                        torch.save(state_dict, str(self.exp_output_dir / f'state_set_{name}_best.dict'))
                torch.save(net.state_dict(), str(self.get_net_path(epoch)))
            if scheduler is not None:
                scheduler.step()

    def retest(self, net, epoch=-1):
        """reference :289-302"""
        epochs = range(self.epochs) if epoch < 0 else [epoch]
        test_sets = self.get_test_sets()
        for epoch in epochs:
            net_path = self.get_net_path(epoch)
            if net_path.exists():
                net.load_state_dict(torch.load(str(net_path)))
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


class TestSet(object):
    """reference data/base_dataset.py:29-37"""

    def __init__(self, name, dset, test_frequency=1):
        self.name = name
        self.dset = dset
        self.test_frequency = test_frequency


class TestSets(list):
    def append(self, name, dset, test_frequency=1):
        super().append(TestSet(name, dset, test_frequency))
