"""ea_harness -- the callers either side of the attention hot path.

Builder-written stand-ins for the reference's two harnesses, so that BASELINE.json's configs 2-5 run
as WHOLE models (eager, autocast, DistributedDataParallel over RCCL) on synthetic data:

  vision.DeiTStack      -- the vit/ classifier: patch stem, 12 pre-norm blocks around
                           AttentionFactory.build_attention, mean-pool head
                           (vit/models/efficient_vit.py:85-119,122-233; state_dict keys match)
  vision.PvTStack       -- the four-stage pyramid (3-4-6-3 blocks, sr_ratio -> attention choice)
                           (vit/models/pvt_legacy.py:66-93,187-268; state_dict keys match)
  sequence.EncoderStack -- a fairseq-free Time x Batch x Channel encoder whose self-attention is the
                           adapter of fairseq/fairseq/modules/efficient_attention.py:107-132
  trainer               -- one synthetic training step (forward, loss, backward, optimizer) eagerly or
                           captured in a hipGraph, single process or one process per GPU

Only the attention layers are this repo's product; everything else here is plain PyTorch plumbing
(library GEMMs / MIOpen convolutions) that exists to drive the path the way its call sites do.
"""
from .vision import DeiTStack, PvTStack, deit_tiny, pvt_b2           # noqa: F401
from .sequence import EncoderStack, wmt_en_de_encoder                # noqa: F401
