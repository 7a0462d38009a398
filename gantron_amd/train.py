"""Training entry point with the reference's CLI, schedule and checkpoint format
(reference train.py:114-527), running on the HIP engine.

    python -m gantron_amd.train -o OUT --wavs_path P [--vesus_path V] [--hparams k=v,...]
                                [--n_gpus N --rank R --group_name G] [-c CKPT [--warm_start]]

Schedule (train.py:296-301,357-360,420-423): iterations 0..5 generator; until `disc_warmp_up`
discriminator only (on the cached generator output); afterwards `g_freq` generator steps followed by
`d_freq` discriminator steps; every 10 000 iterations 100 consecutive discriminator steps.
Checkpoints hold {'iteration','state_dict' (generator only, as in the reference),'g_optimizer',
'g_learning_rate','d_optimizer','d_learning_rate'}.
"""
import argparse
import os
import time
from random import sample

import torch
from torch.utils.data import DataLoader
from torch.utils.data.distributed import DistributedSampler

from . import logger
from .data_utils import DeviceBatches, TextAudioCollate, TextAudioLoader, TextMelCollate, TextMelLoader
from .distributed import apply_gradient_allreduce, init_distributed, reduce_tensor
from .hparams import HParams
from .loss_function import Tacotron2Loss
from .model import Discriminator, LinearDiscriminator, Tacotron2
from .optim import FlatAdam
from .train_steps import discriminator_step, generator_step, gradient_penalty  # noqa: F401 (train.py:32)


def prepare_dataloaders(hparams, wavs_path):
    """reference train.py:94-111.  Raw-audio training batches go through the device pipeline
    (data_utils.DeviceBatches: workers decode on the CPU, one batched STFT-mel per batch on the GPU,
    double-buffered); precomputed mels (`load_mel_from_disk`) keep the reference's loader."""
    valset = TextMelLoader(hparams.validation_files, hparams, wavs_path)
    collate_fn = TextMelCollate(hparams.n_frames_per_step)
    if hparams.load_mel_from_disk:
        trainset, train_collate = TextMelLoader(hparams.training_files, hparams, wavs_path), collate_fn
    else:
        trainset, train_collate = TextAudioLoader(hparams.training_files, hparams, wavs_path), TextAudioCollate()
    sampler = DistributedSampler(trainset) if hparams.distributed_run else None
    # one loader process, as in the reference (train.py:107-110), but kept alive across epochs: re-spawning it (and
    # re-importing torch in it) at every epoch start costs seconds, i.e. percents of an LJSpeech epoch at this step rate
    train_loader = DataLoader(trainset, num_workers=1, shuffle=sampler is None, sampler=sampler,
                              batch_size=hparams.batch_size, pin_memory=True, drop_last=True, collate_fn=train_collate,
                              persistent_workers=True)
    if not hparams.load_mel_from_disk:
        from .stft import TacotronSTFT
        train_loader = DeviceBatches(train_loader, TacotronSTFT(*trainset._stft_args).cuda())
    return train_loader, valset, collate_fn


def load_model(hparams):
    generator = Tacotron2(hparams).cuda()
    discriminator = (LinearDiscriminator(hparams) if hparams.discriminator_type == 'linear' else Discriminator(hparams)).cuda()
    if hparams.distributed_run:
        generator = apply_gradient_allreduce(generator)
        discriminator = apply_gradient_allreduce(discriminator)
    return generator, discriminator


def warm_start_model(checkpoint_path, model, ignore_layers):
    assert os.path.isfile(checkpoint_path)
    model_dict = torch.load(checkpoint_path, map_location='cpu')['state_dict']
    if len(ignore_layers) > 0:
        kept = {k: v for k, v in model_dict.items() if k not in ignore_layers}
        model_dict = model.state_dict()
        model_dict.update(kept)
    model.load_state_dict(model_dict)
    return model


def load_checkpoint(checkpoint_path, model, g_optimizer, d_optimizer):
    assert os.path.isfile(checkpoint_path)
    ck = torch.load(checkpoint_path, map_location='cpu')
    model.load_state_dict(ck['state_dict'])
    g_optimizer.load_state_dict(ck['g_optimizer'])
    d_optimizer.load_state_dict(ck['d_optimizer'])
    return model, g_optimizer, d_optimizer, ck['g_learning_rate'], ck['d_learning_rate'], ck['iteration']


def save_checkpoint(model, g_optimizer, g_learning_rate, d_optimizer, d_learning_rate, iteration, filepath):
    torch.save({'iteration': iteration, 'state_dict': model.state_dict(), 'g_optimizer': g_optimizer.state_dict(),
                'g_learning_rate': g_learning_rate, 'd_optimizer': d_optimizer.state_dict(),
                'd_learning_rate': d_learning_rate}, filepath)


def validate(model, criterion, valset, iteration, batch_size, n_gpus, collate_fn, distributed_run, rank, hparams):
    """reference train.py:169-208 (eval mode: BN running statistics; the Prenet dropout stays on)."""
    model.eval()
    with torch.no_grad():
        sampler = DistributedSampler(valset) if distributed_run else None
        loader = DataLoader(valset, sampler=sampler, num_workers=0, shuffle=False, batch_size=batch_size,
                            pin_memory=False, collate_fn=collate_fn)
        tot = [0.0, 0.0, 0.0]
        n = 0
        for n, batch in enumerate(loader, 1):
            x, y = model.parse_batch(batch)
            y_pred = model(x)
            losses = criterion(y_pred, y, x[1], x[-1])
            for j, l in enumerate(losses):
                tot[j] += (reduce_tensor(l.data, n_gpus) if distributed_run else l).item()
        mel, gate, attn = (t / max(n, 1) for t in tot)
    if iteration > hparams.attn_steps:
        attn = 0
    model.train()
    if rank == 0:
        print(f'{iteration} Validation mel loss {mel} gate loss {gate}')
        logger.log_validation(mel, gate, attn, iteration)
    return mel + gate


def train(output_directory, checkpoint_path, warm_start, n_gpus, rank, group_name, hparams, wavs_path):
    if hparams.distributed_run:
        init_distributed(hparams, n_gpus, rank, group_name)
    if rank == 0 and not os.path.isdir(output_directory):
        os.makedirs(output_directory)
        os.chmod(output_directory, 0o775)
    torch.manual_seed(hparams.seed)
    torch.cuda.manual_seed(hparams.seed)
    generator, discriminator = load_model(hparams)
    g_learning_rate, d_learning_rate = hparams.g_learning_rate, hparams.d_learning_rate
    g_optimizer = FlatAdam(generator, lr=g_learning_rate, weight_decay=hparams.weight_decay)
    d_optimizer = FlatAdam(discriminator, lr=d_learning_rate, weight_decay=hparams.weight_decay)
    criterion = Tacotron2Loss()
    train_loader, valset, collate_fn = prepare_dataloaders(hparams, wavs_path)

    iteration, epoch_offset = 0, 0
    if checkpoint_path is not None:
        if warm_start:
            generator = warm_start_model(checkpoint_path, generator, hparams.ignore_layers)
        else:
            generator, g_optimizer, d_optimizer, g_lr, d_lr, iteration = load_checkpoint(
                checkpoint_path, generator, g_optimizer, d_optimizer)
            if hparams.use_saved_learning_rate:
                g_learning_rate, d_learning_rate = g_lr, d_lr
            iteration += 1
            epoch_offset = max(0, int(iteration / len(train_loader)))
    generator.train()
    discriminator.train()

    gen_times, disc_times = 1, 0
    generated_mel_list = []
    n_epochs = hparams.epochs
    if hparams.iterations is not None and hparams.iterations > 0:
        n_epochs = int(hparams.iterations / len(train_loader)) + 1
    iter_rep, gen_warm = 10000, 5
    best_val_loss, best_path, prev_check, prev_val = float('inf'), None, None, float('inf')

    def checkpoint_and_validate():
        nonlocal best_val_loss, best_path, prev_check, prev_val
        from .engine import check_handoffs
        check_handoffs()              # a checkpoint must not be written from steps whose in-launch hand-offs gave up
        val_loss = validate(generator, criterion, valset, iteration, hparams.batch_size, n_gpus, collate_fn,
                            hparams.distributed_run, rank, hparams)
        if rank == 0:
            path = output_directory + f'/iter={iteration}_val-loss={round(val_loss, 6)}.ckpt'
            save_checkpoint(generator, g_optimizer, g_learning_rate, d_optimizer, d_learning_rate, iteration, path)
            if prev_check is not None and val_loss < prev_val and os.path.exists(prev_check):
                os.remove(prev_check)
            if val_loss < best_val_loss:
                if best_path is not None and os.path.exists(best_path):
                    os.remove(best_path)
                best_val_loss, best_path = val_loss, path
            prev_check = path
        prev_val = val_loss

    for epoch in range(epoch_offset, n_epochs):
        for batch in train_loader:
            start = time.perf_counter()
            do_disc = iteration >= iter_rep and iteration - iter_rep * int(iteration / iter_rep) < 100
            if iteration > gen_warm and (disc_times > 0 or iteration < hparams.disc_warmp_up or do_disc):
                if len(generated_mel_list) > disc_times - 1:
                    generated = generated_mel_list[disc_times - 1]
                if iteration < hparams.disc_warmp_up:
                    generated = sample(generated_mel_list, 1)[0]
                losses = discriminator_step(generator, discriminator, d_optimizer, batch, generated, hparams,
                                            lr=d_learning_rate)
                total = reduce_tensor(losses['total'], n_gpus) if hparams.distributed_run else losses['total']
                if rank == 0:
                    logger.log_values(iteration, discriminator_loss=total, real_loss=losses['real'],
                                      fake_loss=losses['fake'], discriminator_learning_rate=d_learning_rate,
                                      discriminator_duration=time.perf_counter() - start)
                disc_times += 1
                if disc_times > hparams.d_freq and iteration >= hparams.disc_warmp_up:
                    disc_times, gen_times = 0, 1
            else:
                losses, generated = generator_step(generator, discriminator, g_optimizer, criterion, batch, hparams,
                                                   iteration=iteration, lr=g_learning_rate)
                generated_mel_list.append(generated)
                if len(generated_mel_list) > hparams.d_freq:
                    generated_mel_list.pop(0)
                total = reduce_tensor(losses['total'], n_gpus) if hparams.distributed_run else losses['total']
                if rank == 0:
                    logger.log_values(iteration, generator_loss=total, adversarial_loss=losses['adv'],
                                      mel_loss=losses['mel'], gate_loss=losses['gate'],
                                      grad_norm=losses['grad_sumsq'].sqrt() if losses['grad_sumsq'] is not None else None,
                                      generator_learning_rate=g_learning_rate,
                                      generation_duration=time.perf_counter() - start,
                                      attention_loss=losses['attn'] if iteration < hparams.attn_steps else None)
                gen_times += 1
                if gen_times > hparams.g_freq and hparams.d_freq > 0:
                    gen_times, disc_times = 0, 1
            if rank == 0:
                # the logger has just read this step's losses (a synchronisation): the cheapest place to find out that an
                # in-launch hand-off gave up.  The optimizer step of such an iteration was not applied on ANY rank
                # (gt_adam_step_guarded; the flag is MAX-reduced with the gradients); the other ranks find out at the next
                # checkpoint at the latest.
                from .engine import check_handoffs
                check_handoffs()
            iteration += 1
            if iteration % hparams.iters_per_checkpoint == 0:
                checkpoint_and_validate()
            if hparams.reduce_lr_steps_every > 0 and iteration % hparams.reduce_lr_steps_every == 0:
                g_learning_rate /= 2
                d_learning_rate /= 2
            if hparams.iterations is not None and iteration >= hparams.iterations:
                checkpoint_and_validate()
                return


def main():
    p = argparse.ArgumentParser()
    p.add_argument('-o', '--output_directory', type=str, default='output')
    p.add_argument('-c', '--checkpoint_path', type=str, default=None)
    p.add_argument('--waveglow_path', type=str, default=None, help='accepted for CLI compatibility; unused')
    p.add_argument('--vesus_path', type=str, default=None)
    p.add_argument('--warm_start', action='store_true')
    p.add_argument('--n_gpus', type=int, default=1)
    p.add_argument('--rank', type=int, default=0)
    p.add_argument('--group_name', type=str, default='group_name')
    p.add_argument('--hparams', type=str, required=False)
    p.add_argument('--wavs_path', type=str, required=True)
    p.add_argument('--resume', type=str, default='')
    p.add_argument('--notes', type=str, default='')
    p.add_argument('--real', type=int, default=1)
    p.add_argument('--attn_steps', type=int, required=False)
    args = p.parse_args()
    hparams = HParams(args.hparams)
    hparams.add_params(args)
    if not hparams.use_noise:
        hparams.noise_size = 0
    if hparams.d_freq == 0:
        hparams.disc_warmp_up = 0
    hparams.distributed_run = hparams.distributed_run or args.n_gpus > 1
    train(args.output_directory, args.checkpoint_path, args.warm_start, args.n_gpus, args.rank, args.group_name,
          hparams, args.wavs_path)


if __name__ == '__main__':
    main()
