"""ESRGAN trainer on MI355X -- interface of torchsr/esrgan/trainer.py:39-560.

Same skeleton as the SRGAN trainer (shared host loop, checkpoints, PSNR test, hipGraph capture,
flat Adam, bucketed RCCL gradient exchange); differences follow the reference:

* pre-training minimises L1 instead of MSE (esrgan/trainer.py:163,385);
* relativistic-average GAN losses on logits, ``BCEWithLogits(real - mean(fake), 1)`` etc.
  (:451-453,468), ``disc_loss = (real + fake) / 2``;
* ``gen_loss = 0.01 * L1 + 1 * VGG + 0.005 * adversarial`` (:469); the reference's second generator forward
  (:462) recomputes the first bit for bit and is not repeated here;
* BOTH phases sit inside ``amp.autocast`` (:384,446,461), so without ``--disable-amp`` the generator,
  the discriminator and VGG19 all multiply bf16-rounded operands with fp32 accumulation in both
  phases (``amp_phases``; BASELINE config 4).  ``--disable-amp`` gives exact fp32.
"""
import torch

from .. import functional as F
from ..layers import no_weight_grad
from ..srgan.trainer import SRGANTrainer
from .discriminator import Discriminator
from .generator import Generator


class ESRGANTrainer(SRGANTrainer):
    phase_prefix = 'esrgan'
    # three gradients meet at the generator's output (pixel, perceptual, adversarial): summing them in another order than autograd
    # moves last bits, and the backward on the side stream bought nothing here (10.14 vs 10.13 ms): forward only
    deep_overlap = False
    generator_cls = Generator
    discriminator_cls = Discriminator
    amp_phases = ('psnr', 'gan')          # esrgan/trainer.py:384 and :446,461
    gen_tail_bucket = 'upsample1.weight'  # data parallel: upsample1/2, conv3, conv4 gradients complete first

    def _initialize_loss(self) -> None:
        """esrgan/trainer.py:159-165."""
        super()._initialize_loss()
        self.l1_loss = F.l1_loss
        self.pixel_loss = F.l1_loss
        self.bce_loss = F.bce_with_logits

    def _phase_disc(self) -> None:
        """esrgan/trainer.py:444-455 (optimizer step is issued by ``_phase_gen`` after the all-reduce)."""
        self._phase_disc_gen()
        self._phase_disc_loss()

    def _phase_disc_gen(self) -> None:
        # the three networks exchange NHWC tensors (as in the SRGAN trainer): the batch is converted once, the super-resolved
        # image never goes through the NCHW module boundary
        with torch.no_grad():
            low4 = F.to_nhwc(self._static['low_res'], 4)
            self._high4 = F.to_nhwc(self._static['high_res'], 4)
        self.disc_optimizer.zero_grad()                                          # :444
        # The reference runs the generator twice per step (:447 and again at :462) on the same input with the same
        # weights -- only the discriminator is updated in between, and RRDBNet has no BatchNorm, dropout or other
        # state -- so the second forward reproduces the first bit for bit.  It is run once, with its graph kept for
        # the generator update (as the reference's own SRGAN loop does, srgan/trainer.py:444,455-456): 587 GFLOP and
        # ~700 launches per step that change no result.
        self._super_res = self.generator.forward_nhwc(low4)                      # :447 (and :462)

    def _phase_disc_loss(self) -> None:
        # :448-453 -- D(real), D(fake) as one batch; both relativistic terms and their mean in the head's launch
        disc_loss, _ = self.discriminator.pair_loss_nhwc(self._high4, self._super_res.detach())
        self._backward(disc_loss)                                                # :455
        self._losses['gan/disc-loss'] = disc_loss.detach()

    def _phase_content(self) -> None:
        """esrgan/trainer.py:459-467: pixel and perceptual terms (the generator output of ``_phase_disc`` is reused)."""
        self.gen_optimizer.zero_grad()                                           # :459
        # :466 -- the 4th NHWC channel is zero on both sides: the mean runs over the 3 real ones
        pixel = self.l1_loss(self._super_res, self._high4, count=self._high4.numel() // 4 * 3)
        content = self.vgg_loss.forward_nhwc(self._super_res, self._high4)       # :467
        self._content = F.axpby(pixel, content, 0.01, 1.0)
        self._losses['gan/pixel-loss'] = pixel.detach()
        self._losses['gan/content-loss'] = content.detach()

    def _phase_gen(self) -> None:
        """esrgan/trainer.py:456,463-480: D update, relativistic adversarial term, G backward."""
        self.disc_optimizer.step()                                               # :456
        with no_weight_grad():
            with torch.no_grad():  # mean(real_output) carries no gradient to the generator
                real_mean = F.mean(self.discriminator.forward_nhwc(self._high4))  # :463
            # :464,468-469 -- gen_loss = 0.01 * pixel + content + 0.005 * BCEWithLogits(D(fake) - mean(D(real)), 1)
            gen_loss, aux = self.discriminator.adversarial_loss_nhwc(self._super_res, real_mean, self._content, 0.005)
        self._backward(gen_loss)                                                 # :480
        self._losses['gan/adversarial-loss'] = aux[1]
        self._losses['gan/train-loss'] = gen_loss.detach()
        self._super_res = self._content = self._high4 = None
