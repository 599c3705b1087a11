"""Nearest-neighbour retrieval in embedding space.

The reference's notebook export searches the closest audio representations with ``torch.cdist``
(/root/reference/evaluate_audio_representations.py:202-231; the file itself is stale and does not import).
Here: embed a bank of rendered voices with a (frozen) ``VicregAudioParams``, then for query audio return the
indices / distances of the k closest bank items.  ``torch.cdist`` is a plain library GEMM on ROCm.
"""
import torch


@torch.no_grad()
def embed_audio(model, audio):
    """audio [B, T] -> representation [B, dim] with the audio backbone (no projector), eval mode."""
    was_training = model.training
    model.eval()
    try:
        return model.vicreg.backbone_audio(audio.unsqueeze(1))
    finally:
        model.train(was_training)


@torch.no_grad()
def build_bank(model, batch_indices):
    """Render the given voice batches and embed them -> (embeddings [N, dim], params [N, 78])."""
    embs, params = [], []
    for idx in batch_indices:
        audio, p, _ = model.voice(int(idx))
        embs.append(embed_audio(model, audio))
        params.append(p)
    return torch.cat(embs), torch.cat(params)


@torch.no_grad()
def nearest(queries, bank, k=1):
    """-> (distances [Q, k], indices [Q, k]) of the k nearest bank rows (Euclidean, as torch.cdist)."""
    d = torch.cdist(queries, bank)
    dist, idx = torch.topk(d, k, dim=1, largest=False)
    return dist, idx
