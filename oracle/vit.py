"""A10: CLIP ViT-B/32 visual tower forward, fp32 torch-CPU ops (test infrastructure).

Restates openai/CLIP ``clip/model.py`` ``VisionTransformer.forward`` (``clip==1.0``,
requirements/environment.yaml:98; not vendored) as called at hub/compressor.py:93, from
the recipe in SURVEY.md section 8(a) row A10 and section 9.3.  Takes the OpenAI state-dict
layout with the ``visual.`` prefix stripped.  QuickGELU, LayerNorm eps 1e-5, pre-LN
residual blocks, 12 heads of 64, class token first, ``ln_post`` on the class token
only, then ``@ proj``.
"""
import torch
import torch.nn.functional as F

WIDTH, LAYERS, HEADS, PATCH, RES, OUT = 768, 12, 12, 32, 224, 512


def vit_b32_forward(sd, images_nchw, weights_rounded_to_fp16=True, fp16_storage=False):
    """images_nchw: [B,3,224,224] float -> z [B,512] fp32.

    With ``weights_rounded_to_fp16`` the parameters are first rounded to fp16 (what
    ``clip.load`` keeps on a GPU) and then used in fp32 arithmetic, so the comparison
    against the HIP path isolates arithmetic error from weight quantisation.

    ``fp16_storage``: still fp32 arithmetic, but the LayerNorm outputs, qkv, the attention
    output, the MLP hidden activation and the final embedding are rounded to fp16 -- the places
    where the HIP tower stores fp16 (the reference's own fp16 CLIP rounds at least as often).
    The difference to the plain fp32 result is the error inherent to fp16 activations; a kernel
    should add nothing on top of it.
    """
    def w(name):
        t = sd[name]
        t = t.half().float() if weights_rounded_to_fp16 else t.float()
        return t

    def r(t):
        return t.half().float() if fp16_storage else t

    x = images_nchw.float()
    B = x.shape[0]
    x = F.conv2d(x, w("conv1.weight"), stride=PATCH)             # [B,768,7,7]
    x = x.reshape(B, WIDTH, -1).permute(0, 2, 1)                # [B,49,768]
    cls = w("class_embedding").reshape(1, 1, WIDTH).expand(B, 1, WIDTH)
    x = torch.cat([cls, x], dim=1) + w("positional_embedding")  # [B,50,768]
    x = F.layer_norm(x, (WIDTH,), w("ln_pre.weight"), w("ln_pre.bias"), 1e-5)
    for l in range(LAYERS):
        p = "transformer.resblocks.%d." % l
        h = r(F.layer_norm(x, (WIDTH,), w(p + "ln_1.weight"), w(p + "ln_1.bias"), 1e-5))
        qkv = r(h @ w(p + "attn.in_proj_weight").t() + w(p + "attn.in_proj_bias"))
        q, k, v = qkv.split(WIDTH, dim=-1)
        hd = WIDTH // HEADS
        q = q.reshape(B, -1, HEADS, hd).transpose(1, 2)
        k = k.reshape(B, -1, HEADS, hd).transpose(1, 2)
        v = v.reshape(B, -1, HEADS, hd).transpose(1, 2)
        att = torch.softmax((q @ k.transpose(-1, -2)) * (hd ** -0.5), dim=-1)
        o = r((att @ v).transpose(1, 2).reshape(B, -1, WIDTH))
        x = x + o @ w(p + "attn.out_proj.weight").t() + w(p + "attn.out_proj.bias")
        h = r(F.layer_norm(x, (WIDTH,), w(p + "ln_2.weight"), w(p + "ln_2.bias"), 1e-5))
        h = h @ w(p + "mlp.c_fc.weight").t() + w(p + "mlp.c_fc.bias")
        h = r(h * torch.sigmoid(1.702 * h))
        x = x + h @ w(p + "mlp.c_proj.weight").t() + w(p + "mlp.c_proj.bias")
    x = r(F.layer_norm(x[:, 0, :], (WIDTH,), w("ln_post.weight"), w("ln_post.bias"), 1e-5))
    return r(x @ w("proj"))
