"""Epoch loop, learning-rate schedule and checkpoints around the MI355X training step.

Mirrors the reference's trainer (egs/wsj0-mix/common/src/driver.py:20-226 `TrainerBase`, and the Conv-TasNet
recipe's `AdhocTrainer`, egs/wsj0-mix/conv-tasnet/src/adhoc_driver.py:7-44): same constructor arguments, same
`args` attributes, same per-epoch decisions and the same checkpoint dictionary, so that a `.pth` written by either
side resumes on the other.  What differs is underneath: the optimizer step is the fused clip + Adam kernel of
`FlatTrainer` on flat buffers, the `torch.optim.Adam` object the caller passes in supplies the hyper-parameters,
carries the learning rate (the schedule halves `param_groups[...]['lr']` exactly as the reference does) and gives
the checkpoint its `optim_dict` layout.
"""
import os
import time
from collections import OrderedDict

import torch
import torch.distributed as dist

from .dataset import save_wav
from .trainer import FlatTrainer

BITS_PER_SAMPLE_WSJ0 = 16


class EpochControl:
    """The reference's per-epoch bookkeeping (driver.py:95-115): best loss, epochs without improvement, learning-rate
    halving after 3 of them and, with `early_stop`, the end of training after 10 (adhoc_driver.py:21-37 has no
    early stop)."""

    def __init__(self, early_stop=True):
        self.best_loss = float('infinity')
        self.prev_loss = float('infinity')
        self.no_improvement = 0
        self.early_stop = early_stop

    def update(self, valid_loss):
        """Returns (save_best, halve_lr, stop) for the epoch that produced `valid_loss`."""
        save_best = halve = stop = False
        if valid_loss < self.best_loss:
            self.best_loss = valid_loss
            self.no_improvement = 0
            save_best = True
        else:
            if valid_loss >= self.prev_loss:
                self.no_improvement += 1
                if self.early_stop and self.no_improvement >= 10:
                    stop = True
                elif self.no_improvement >= 3:
                    halve = True
            else:
                self.no_improvement = 0
        if not stop:                      # the reference leaves the loop before it records the loss (driver.py:103-105)
            self.prev_loss = valid_loss
        return save_best, halve, stop


def _adam_group(optimizer):
    if not isinstance(optimizer, torch.optim.Adam) or type(optimizer) is not torch.optim.Adam:
        raise NotImplementedError("the HIP training step implements torch.optim.Adam (given {})".format(
            type(optimizer).__name__))
    if len(optimizer.param_groups) != 1:
        raise NotImplementedError("one parameter group expected")
    g = optimizer.param_groups[0]
    if g.get('weight_decay', 0) or g.get('amsgrad', False) or g.get('maximize', False):
        raise NotImplementedError("Adam with weight decay / amsgrad / maximize is outside the fused step")
    return g


def export_adam_state(optimizer, flat_trainer):
    """`optimizer.state_dict()` with the per-parameter state taken from the flat moment buffers: the layout of
    torch.optim.Adam (state[i] = {step, exp_avg, exp_avg_sq}, i = position in param_groups[0]['params'])."""
    sd = optimizer.state_dict()
    state = {}
    if flat_trainer.step_count > 0:
        views = flat_trainer.moment_views()
        for i, p in enumerate(optimizer.param_groups[0]['params']):
            m, v = views[id(p)]
            state[i] = {'step': torch.tensor(float(flat_trainer.step_count)), 'exp_avg': m.detach().cpu().clone(),
                        'exp_avg_sq': v.detach().cpu().clone()}
    sd['state'] = state
    return sd


def import_adam_state(optimizer, flat_trainer, optim_dict):
    """Inverse of export_adam_state; also accepts a checkpoint written by the reference (driver.py:216), whose
    `step` is an int under torch 1.10 and a tensor under later versions."""
    group = optim_dict['param_groups'][0]
    g = optimizer.param_groups[0]
    for key in ('lr', 'betas', 'eps'):
        if key in group:
            g[key] = group[key]
    state = optim_dict.get('state', {})
    views = flat_trainer.moment_views()
    steps = set()
    with torch.no_grad():
        for i, p in enumerate(g['params']):
            st = state.get(i, state.get(str(i)))
            m, v = views[id(p)]
            if st is None:
                m.zero_(); v.zero_()
                continue
            m.copy_(st['exp_avg'].to(m.device).view_as(m))
            v.copy_(st['exp_avg_sq'].to(v.device).view_as(v))
            steps.add(int(float(st['step'])))
    if len(steps) > 1:
        raise ValueError("parameters with different Adam step counts {} cannot share the fused step".format(sorted(steps)))
    flat_trainer.step_count = steps.pop() if steps else 0
    flat_trainer.betas, flat_trainer.eps = tuple(g['betas']), g['eps']


def allreduce_host_safe(t, group=None):
    """Sum-all-reduce of a small device tensor in place; through a host copy when the process group is gloo (several
    ranks rehearsing on one GPU), directly on RCCL otherwise."""
    if t.is_cuda and dist.get_backend(group) == "gloo":
        host = t.cpu()
        dist.all_reduce(host, group=group)
        t.copy_(host)
    else:
        dist.all_reduce(t, group=group)
    return t


class TrainerBase:
    """egs/wsj0-mix/common/src/driver.py:20-226.  `args` needs: sample_rate, n_sources, max_norm, model_dir, loss_dir,
    sample_dir, epochs, use_cuda, continue_from, overwrite.  Extra keyword arguments go to FlatTrainer
    (n_streams, use_graph, process_group, n_buckets)."""

    early_stop = True

    def __init__(self, model, loader, pit_criterion, optimizer, args, **flat_kwargs):
        self.train_loader, self.valid_loader = loader['train'], loader['valid']
        self.model = model
        self.pit_criterion = pit_criterion
        self.optimizer = optimizer
        group = _adam_group(optimizer)
        max_norm = args.max_norm if args.max_norm else None
        self.flat = FlatTrainer(model, pit_criterion, lr=group['lr'], max_norm=max_norm, betas=tuple(group['betas']),
                                eps=group['eps'], **flat_kwargs)
        self.rank = dist.get_rank(self.flat.group) if self.flat.world > 1 else 0
        self._reset(args)

    def _reset(self, args):
        self.sample_rate = args.sample_rate
        self.n_sources = args.n_sources
        self.max_norm = args.max_norm
        self.model_dir, self.loss_dir, self.sample_dir = args.model_dir, args.loss_dir, args.sample_dir
        if self.rank == 0:
            os.makedirs(self.model_dir, exist_ok=True)
            os.makedirs(self.loss_dir, exist_ok=True)
            os.makedirs(self.sample_dir, exist_ok=True)
        self.epochs = args.epochs
        self.train_loss = torch.empty(self.epochs)
        self.valid_loss = torch.empty(self.epochs)
        self.use_cuda = args.use_cuda
        if not self.use_cuda:
            raise ValueError("this trainer runs on the GPU only (use_cuda must be set)")
        self.control = EpochControl(early_stop=self.early_stop)

        if getattr(args, 'continue_from', None):
            config = torch.load(args.continue_from, map_location="cpu", weights_only=True)
            self.start_epoch = config['epoch']
            self.train_loss[:self.start_epoch] = config['train_loss'][:self.start_epoch]
            self.valid_loss[:self.start_epoch] = config['valid_loss'][:self.start_epoch]
            self.control.best_loss = config['best_loss']
            self.control.prev_loss = self.valid_loss[self.start_epoch - 1].item()
            self.control.no_improvement = config['no_improvement']
            self.model.load_state_dict(config['state_dict'])      # parameters are views of the flat buffer: in place
            import_adam_state(self.optimizer, self.flat, config['optim_dict'])
        else:
            model_path = os.path.join(self.model_dir, "best.pth")
            if os.path.exists(model_path):
                if getattr(args, 'overwrite', False):
                    print("Overwrite models.")
                else:
                    raise ValueError("{} already exists. If you continue to run, set --overwrite to be True.".format(
                        model_path))
            self.start_epoch = 0

    # the reference keeps these three on the trainer itself (driver.py:80-82); checkpoints store them
    @property
    def best_loss(self):
        return self.control.best_loss

    @property
    def prev_loss(self):
        return self.control.prev_loss

    @property
    def no_improvement(self):
        return self.control.no_improvement

    def run(self):
        for epoch in range(self.start_epoch, self.epochs):
            start = time.time()
            train_loss, valid_loss = self.run_one_epoch(epoch)
            end = time.time()
            if self.rank == 0:
                print("[Epoch {}/{}] loss (train): {:.5f}, loss (valid): {:.5f}, {:.3f} [sec]".format(
                    epoch + 1, self.epochs, train_loss, valid_loss, end - start), flush=True)
            self.train_loss[epoch] = train_loss
            self.valid_loss[epoch] = valid_loss

            save_best, halve, stop = self.control.update(valid_loss)
            if save_best:
                self.save_model(epoch, os.path.join(self.model_dir, "best.pth"))
            if stop:
                if self.rank == 0:
                    print("Stop training")
                break
            if halve:
                for param_group in self.optimizer.param_groups:
                    prev_lr = param_group['lr']
                    lr = 0.5 * prev_lr
                    if self.rank == 0:
                        print("Learning rate: {} -> {}".format(prev_lr, lr))
                    param_group['lr'] = lr
            self.save_model(epoch, os.path.join(self.model_dir, "last.pth"))
            if self.rank == 0:                # driver.py:120-121
                try:
                    from .utils.utils import draw_loss_curve
                    draw_loss_curve(train_loss=self.train_loss[:epoch + 1], valid_loss=self.valid_loss[:epoch + 1],
                                    save_path=os.path.join(self.loss_dir, "loss.png"))
                except ImportError:           # no plotting library on this node: the numbers are in the checkpoint
                    pass

    def run_one_epoch(self, epoch):
        train_loss = self.run_one_epoch_train(epoch)
        valid_loss = self.run_one_epoch_eval(epoch)
        return train_loss, valid_loss

    def run_one_epoch_train(self, epoch):
        """driver.py:132-164.  The loss of every step is added up on the device in float64 -- the value the
        reference gets from adding `loss.item()` to a Python float -- and read once per epoch, so the steps run
        without a host synchronisation (every 100th step is read for the progress line, as in the reference)."""
        self.model.train()
        dev = self.flat.flat_p.device
        total = torch.zeros((), device=dev, dtype=torch.float64)
        n_train_batch = len(self.train_loader)
        for idx, (mixture, sources) in enumerate(self.train_loader):
            mixture = mixture.to(dev, non_blocking=True)
            sources = sources.to(dev, non_blocking=True)
            self.flat.lr = self.optimizer.param_groups[0]['lr']
            loss, _ = self.flat.step(mixture, sources)
            total += loss.double()
            if (idx + 1) % 100 == 0 and self.rank == 0:
                print("[Epoch {}/{}] iter {}/{} loss: {:.5f}".format(epoch + 1, self.epochs, idx + 1, n_train_batch,
                                                                     loss.item()), flush=True)
        if self.flat.world > 1:               # every rank saw an equal shard of every batch: mean of the shard means
            allreduce_host_safe(total, self.flat.group)
            total /= self.flat.world
        return total.item() / n_train_batch

    def run_one_epoch_eval(self, epoch):
        """driver.py:166-206: PIT loss of every validation utterance (batch_mean=False), summed, over the number of
        utterances.  The reference divides by len(valid_loader.dataset), which in its single process is the number of
        utterances the loop saw; here that number is counted, so that with several ranks -- each given its own part of
        the set by a disjoint dataset or by a DistributedSampler over a shared one, padding duplicates included -- the
        all-reduced sum is divided by what was actually evaluated."""
        self.model.eval()
        dev = self.flat.flat_p.device
        total = torch.zeros((), device=dev, dtype=torch.float64)
        n_seen = 0
        with torch.no_grad():
            for idx, batch in enumerate(self.valid_loader):
                mixture, sources = batch[0].to(dev), batch[1].to(dev)
                output = self.model(mixture)
                loss, _ = self.pit_criterion(output, sources, batch_mean=False)
                total += loss.sum(dim=0).double()
                n_seen += loss.numel()
                if idx < 5 and self.rank == 0 and len(batch) > 2:
                    self.save_samples(epoch, batch[2][0], mixture[0].squeeze(dim=0).cpu(), output[0].cpu())
        n = torch.tensor(float(n_seen), device=dev, dtype=torch.float64)
        if self.flat.world > 1:               # every rank validates its own part of the set
            pair = torch.stack([total, n])
            allreduce_host_safe(pair, self.flat.group)
            total, n = pair[0], pair[1]
        if n.item() == 0:
            raise ValueError("the validation loader produced no utterances")
        return (total / n).item()

    def save_samples(self, epoch, segment_ID, mixture, estimated_sources):
        """driver.py:185-202: the first five validation mixtures and their estimates, peak-normalised, as 16-bit
        wave files under sample_dir/<segment ID>/."""
        save_dir = os.path.join(self.sample_dir, segment_ID)
        os.makedirs(save_dir, exist_ok=True)
        mixture = mixture / torch.abs(mixture).max()
        save_wav(os.path.join(save_dir, "mixture.wav"), mixture.unsqueeze(dim=0) if mixture.dim() == 1 else mixture,
                 self.sample_rate, bits_per_sample=BITS_PER_SAMPLE_WSJ0)
        for source_idx, estimated_source in enumerate(estimated_sources):
            estimated_source = estimated_source / torch.abs(estimated_source).max()
            signal = estimated_source.unsqueeze(dim=0) if estimated_source.dim() == 1 else estimated_source
            save_wav(os.path.join(save_dir, "epoch{}-{}.wav".format(epoch + 1, source_idx + 1)), signal, self.sample_rate,
                     bits_per_sample=BITS_PER_SAMPLE_WSJ0)

    def save_model(self, epoch, model_path='./tmp.pth'):
        """driver.py:208-226: the model's config, its state_dict, the optimizer's state_dict and the loss history."""
        if self.rank != 0:
            return
        config = self.model.get_config()
        config['state_dict'] = OrderedDict((k, v.detach().cpu().clone()) for k, v in self.model.state_dict().items())
        config['optim_dict'] = export_adam_state(self.optimizer, self.flat)
        config['best_loss'] = self.best_loss
        config['no_improvement'] = self.no_improvement
        config['train_loss'] = self.train_loss
        config['valid_loss'] = self.valid_loss
        config['epoch'] = epoch + 1
        torch.save(config, model_path)


class TesterBase:
    """Placeholder for the reference's evaluation driver (egs/wsj0-mix/common/src/driver.py:228-390): it scores with
    mir_eval's BSS metrics and an external PESQ executable, neither of which is part of the accelerated path.  The name
    exists so that recipe modules importing it keep importing; constructing it says what is missing."""

    def __init__(self, *args, **kwargs):
        raise NotImplementedError("evaluation (mir_eval BSS metrics, PESQ subprocess) is outside the MI355X training path; "
                                  "run the reference's tester on a checkpoint written by this trainer instead")


class AdhocTrainer(TrainerBase):
    """egs/wsj0-mix/conv-tasnet/src/adhoc_driver.py:7-44: the same loop without the early stop."""

    early_stop = False
