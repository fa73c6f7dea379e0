"""IP-Adapter image-token projection pinned by the reference's own ImageProjModel (modules/ip_adapter.py:30-47) and the
cond / uncond rule of get_image_embeds (:187-198: uncond = proj(zeros)).  Container only.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_ip_golden.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refstub  # noqa: E402

_refstub.install()
import numpy as np  # noqa: E402
import torch  # noqa: E402

from modules.ip_adapter import ImageProjModel  # noqa: E402  (reference)

if __name__ == "__main__":
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(2024)
    m = ImageProjModel(cross_attention_dim=768, clip_embeddings_dim=1024, clip_extra_context_tokens=4).eval()
    sd = {"proj.weight": torch.randn(4 * 768, 1024, generator=g) * 1024 ** -0.5, "proj.bias": torch.randn(4 * 768, generator=g) * 0.1,
          "norm.weight": 1 + 0.1 * torch.randn(768, generator=g), "norm.bias": 0.1 * torch.randn(768, generator=g)}
    m.load_state_dict(sd)
    emb = torch.randn(2, 1024, generator=g)
    with torch.no_grad():
        tokens = m(emb)
        uncond = m(torch.zeros_like(emb))
    # (the 12 MB projection matrix is not stored: the test re-draws the weights from the same generator sequence)
    np.savez_compressed(os.path.join(HERE, "ip_image_proj.npz"), clip_image_embeds=emb.numpy(), tokens=tokens.numpy(), uncond=uncond.numpy(),
                        weight_seed=2024, weight_checksum=float(sum(v.double().abs().sum() for v in sd.values())))
    print("wrote ip_image_proj.npz", tuple(tokens.shape))
