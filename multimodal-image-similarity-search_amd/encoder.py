"""ClipEncoder — batched CLIP image / text towers on one MI355X through libmmiss.

The arithmetic replaced is `model.get_image_features(**inputs)` / `model.get_text_features(**inputs)`
followed by the L2 normalisation in backend/app/utils.py:77-78,97-98 of the reference. Weights use the
HF state_dict key layout (SURVEY.md §8 a-W), so a local HF checkpoint loads unchanged.
"""
from __future__ import annotations

import ctypes as C
import threading
from dataclasses import dataclass, asdict
from typing import Dict, Iterable, Optional, Tuple

import numpy as np

from . import _lib


@dataclass
class ClipShape:
    """Shape of both towers; defaults = HF CLIPConfig() = ViT-B/32 (HF:configuration_clip.py:47-54,97-105,160)."""

    v_hidden: int = 768
    v_layers: int = 12
    v_heads: int = 12
    v_mlp: int = 3072
    v_patch: int = 32
    v_image: int = 224
    t_hidden: int = 512
    t_layers: int = 12
    t_heads: int = 8
    t_mlp: int = 2048
    t_vocab: int = 49408
    t_ctx: int = 77
    proj_dim: int = 512
    eos_token_id: int = 49407
    ln_eps: float = 1e-5

    @classmethod
    def from_any(cls, obj) -> "ClipShape":
        """Accepts another dataclass / dict with the same field names (e.g. the oracle's ClipShape)."""
        if isinstance(obj, cls):
            return obj
        d = obj if isinstance(obj, dict) else asdict(obj)
        return cls(**{k: d[k] for k in cls.__dataclass_fields__})

    @classmethod
    def from_hf_config(cls, cfg) -> "ClipShape":
        """From a transformers CLIPConfig (or the dict of a checkpoint's config.json): the reference builds one at
        utils.py:41-42. A saved config.json may omit every key whose value equals the HF default (`to_diff_dict`), so
        missing keys fall back to CLIPVisionConfig / CLIPTextConfig's defaults (HF:configuration_clip.py:47-54,97-105,160).
        Only `quick_gelu` towers are supported (the kernels fuse x*sigmoid(1.702x), HF:activations.py:117-123): any
        other `hidden_act` raises instead of producing silently different embeddings."""
        d = cfg.to_dict() if hasattr(cfg, "to_dict") else dict(cfg)
        v = dict(d.get("vision_config") or d.get("vision_config_dict") or {})
        t = dict(d.get("text_config") or d.get("text_config_dict") or {})
        vd = dict(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072, patch_size=32,
                  image_size=224, layer_norm_eps=1e-5, hidden_act="quick_gelu")
        td = dict(hidden_size=512, num_hidden_layers=12, num_attention_heads=8, intermediate_size=2048, vocab_size=49408,
                  max_position_embeddings=77, layer_norm_eps=1e-5, hidden_act="quick_gelu", eos_token_id=49407)
        v = {**vd, **{k: x for k, x in v.items() if x is not None}}
        t = {**td, **{k: x for k, x in t.items() if x is not None}}
        for name, c in (("vision_config", v), ("text_config", t)):
            if c["hidden_act"] != "quick_gelu":
                raise ValueError(f"{name}.hidden_act = {c['hidden_act']!r}: only 'quick_gelu' towers are supported")
        if abs(float(v["layer_norm_eps"]) - float(t["layer_norm_eps"])) > 0:
            raise ValueError("vision and text towers with different layer_norm_eps are not supported")
        proj = d.get("projection_dim")
        if proj is None:
            proj = v.get("projection_dim", t.get("projection_dim", 512))
        return cls(v_hidden=int(v["hidden_size"]), v_layers=int(v["num_hidden_layers"]), v_heads=int(v["num_attention_heads"]),
                   v_mlp=int(v["intermediate_size"]), v_patch=int(v["patch_size"]), v_image=int(v["image_size"]),
                   t_hidden=int(t["hidden_size"]), t_layers=int(t["num_hidden_layers"]), t_heads=int(t["num_attention_heads"]),
                   t_mlp=int(t["intermediate_size"]), t_vocab=int(t["vocab_size"]), t_ctx=int(t["max_position_embeddings"]),
                   proj_dim=int(proj), eos_token_id=int(t["eos_token_id"]), ln_eps=float(v["layer_norm_eps"]))

    @property
    def v_tokens(self) -> int:
        return (self.v_image // self.v_patch) ** 2 + 1


VIT_B32 = ClipShape()
LONGCLIP_L14 = ClipShape(v_hidden=1024, v_layers=24, v_heads=16, v_mlp=4096, v_patch=14, v_image=224,
                         t_hidden=768, t_layers=12, t_heads=12, t_mlp=3072, t_ctx=248, proj_dim=768)


def _is_torch(x) -> bool:
    return hasattr(x, "data_ptr")


def random_state_dict(shape: ClipShape, seed: int = 0) -> Dict[str, np.ndarray]:
    """Seeded random-init weights in the HF state_dict layout (no checkpoints exist offline): matrix stds
    follow HF's init table (HF:modeling_clip.py:404-437); biases / LayerNorm affine get small random values so
    every term of the forward is exercised. Used by bench.py, smoke() and the parity tests."""
    s = ClipShape.from_any(shape)
    rng = np.random.Generator(np.random.Philox(seed))
    w: Dict[str, np.ndarray] = {}

    def normal(name, shp, std):
        w[name] = (rng.standard_normal(shp, dtype=np.float32) * np.float32(std)).astype(np.float32)

    def ln(prefix, d):
        w[prefix + ".weight"] = (1.0 + 0.1 * rng.standard_normal(d, dtype=np.float32)).astype(np.float32)
        w[prefix + ".bias"] = (0.1 * rng.standard_normal(d, dtype=np.float32)).astype(np.float32)

    def tower(prefix, d, layers, mlp):
        in_std = d ** -0.5 * (2 * layers) ** -0.5
        out_std = d ** -0.5
        fc_std = (2 * d) ** -0.5
        for i in range(layers):
            p = f"{prefix}.encoder.layers.{i}."
            for nm in ("q_proj", "k_proj", "v_proj"):
                normal(p + f"self_attn.{nm}.weight", (d, d), in_std)
                normal(p + f"self_attn.{nm}.bias", (d,), 0.02)
            normal(p + "self_attn.out_proj.weight", (d, d), out_std)
            normal(p + "self_attn.out_proj.bias", (d,), 0.02)
            ln(p + "layer_norm1", d)
            ln(p + "layer_norm2", d)
            normal(p + "mlp.fc1.weight", (mlp, d), fc_std)
            normal(p + "mlp.fc1.bias", (mlp,), 0.02)
            normal(p + "mlp.fc2.weight", (d, mlp), in_std)
            normal(p + "mlp.fc2.bias", (d,), 0.02)

    normal("vision_model.embeddings.class_embedding", (s.v_hidden,), s.v_hidden ** -0.5)
    normal("vision_model.embeddings.patch_embedding.weight", (s.v_hidden, 3, s.v_patch, s.v_patch), 0.02)
    normal("vision_model.embeddings.position_embedding.weight", (s.v_tokens, s.v_hidden), 0.02)
    ln("vision_model.pre_layrnorm", s.v_hidden)
    tower("vision_model", s.v_hidden, s.v_layers, s.v_mlp)
    ln("vision_model.post_layernorm", s.v_hidden)
    normal("visual_projection.weight", (s.proj_dim, s.v_hidden), s.v_hidden ** -0.5)
    normal("text_model.embeddings.token_embedding.weight", (s.t_vocab, s.t_hidden), 0.02)
    normal("text_model.embeddings.position_embedding.weight", (s.t_ctx, s.t_hidden), 0.02)
    tower("text_model", s.t_hidden, s.t_layers, s.t_mlp)
    ln("text_model.final_layer_norm", s.t_hidden)
    normal("text_projection.weight", (s.proj_dim, s.t_hidden), s.t_hidden ** -0.5)
    return w


def iter_safetensors_f32(path: str):
    """(key, float32 C-contiguous ndarray) for every tensor of a safetensors file. Read through torch because numpy has
    no bfloat16 (`safe_open(framework="np")` refuses bf16 checkpoints)."""
    from safetensors import safe_open

    with safe_open(path, framework="pt", device="cpu") as f:
        for key in f.keys():
            yield key, np.ascontiguousarray(f.get_tensor(key).float().numpy())


def safetensors_shapes(path: str) -> Dict[str, Tuple[int, ...]]:
    """Tensor shapes from the file header only (no tensor data is read)."""
    from safetensors import safe_open

    with safe_open(path, framework="pt", device="cpu") as f:
        return {k: tuple(f.get_slice(k).get_shape()) for k in f.keys()}


class ClipEncoder:
    """One encoder handle = both towers' weights in HBM + workspaces, on one GPU, one HIP stream."""

    def __init__(self, shape: ClipShape = VIT_B32, device: int = 0, max_batch_image: int = 256,
                 max_batch_text: int = 256, precision: str = "bf16"):
        self.shape = ClipShape.from_any(shape)
        self.device = int(device)
        self._lib = _lib.load()
        s = self.shape
        cfg = _lib.ClipConfigStruct(
            C.sizeof(_lib.ClipConfigStruct), s.v_hidden, s.v_layers, s.v_heads, s.v_mlp, s.v_patch, s.v_image,
            s.t_hidden, s.t_layers, s.t_heads, s.t_mlp, s.t_vocab, s.t_ctx, s.proj_dim, s.eos_token_id,
            float(s.ln_eps), int(max_batch_image), int(max_batch_text))
        h = C.c_void_p()
        _lib.check(self._lib.mmiss_encoder_create(C.byref(cfg), self.device, C.byref(h)))
        self._h = h
        self._call_lock = threading.Lock()
        self._finalized = False
        self.precision = "bf16"
        if precision != "bf16":
            self.set_precision(precision)

    def set_precision(self, precision: str) -> None:
        """"bf16" (default): bf16 GEMM operands, f32 accumulation; large calls (>= ~6000 token rows) also keep the residual
        stream in bf16 (1 - cos vs the fp32 oracle 5e-5 instead of 5e-6, 4-5 % faster). "bf16-f32resid": the residual
        stream f32 at every batch size. "fp8": the QKV / FC1 / FC2 projections of the VISION tower on the block-scaled fp8
        matrix cores (e4m3 operands, f32 accumulation; mmiss_encoder_set_precision) - BASELINE configs[4]; the text tower
        stays bf16 under this setting (its fp8 form is outside the tolerance). Bar for all of them: 1 - cos <= 1e-3."""
        code = {"bf16": _lib.MMISS_PREC_BF16, "fp8": _lib.MMISS_PREC_FP8, "bf16-f32resid": _lib.MMISS_PREC_BF16_F32RESID}[precision]
        _lib.check(self._lib.mmiss_encoder_set_precision(self._h, code))
        self.precision = precision

    def set_tower_precision(self, tower: str, precision: str) -> None:
        """One tower ("vision" / "text") to "bf16" or "fp8" (mmiss_encoder_set_tower_precision). fp8 on the text tower is an
        explicit opt-in: it measures 1 - cos = 3-4e-3 against the fp32 oracle, outside the 1e-3 tolerance."""
        t = {"vision": _lib.MMISS_TOWER_VISION, "text": _lib.MMISS_TOWER_TEXT}[tower]
        code = {"bf16": _lib.MMISS_PREC_BF16, "fp8": _lib.MMISS_PREC_FP8}[precision]
        _lib.check(self._lib.mmiss_encoder_set_tower_precision(self._h, t, code))

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, state: Dict[str, "np.ndarray"]) -> Tuple[int, int]:
        """state: HF key -> float32 array (numpy or torch CPU tensor). Returns (used, ignored)."""
        used = ignored = 0
        for key, val in state.items():
            if _is_torch(val):
                val = val.detach().to("cpu").float().contiguous().numpy()
            arr = np.ascontiguousarray(val, dtype=np.float32)
            u = C.c_int(0)
            _lib.check(self._lib.mmiss_encoder_set_weight(self._h, key.encode(), arr.ctypes.data, arr.size, C.byref(u)))
            used += u.value
            ignored += 1 - u.value
        _lib.check(self._lib.mmiss_encoder_finalize(self._h))
        self._finalized = True
        return used, ignored

    def load_safetensors(self, path: str) -> Tuple[int, int]:
        """Load a local HF-layout checkpoint (`model.safetensors`, f32 / f16 / bf16 tensors); no network access is
        attempted. Tensors are widened to f32 and handed over one at a time."""
        used = ignored = 0
        for key, arr in iter_safetensors_f32(path):
            u = C.c_int(0)
            _lib.check(self._lib.mmiss_encoder_set_weight(self._h, key.encode(), arr.ctypes.data, arr.size, C.byref(u)))
            used += u.value
            ignored += 1 - u.value
        _lib.check(self._lib.mmiss_encoder_finalize(self._h))
        self._finalized = True
        return used, ignored

    # ------------------------------------------------------------------ encode
    def _out(self, like, n: int):
        if _is_torch(like) and like.is_cuda:
            import torch

            return torch.empty((n, self.shape.proj_dim), dtype=torch.float32, device=like.device)
        return np.empty((n, self.shape.proj_dim), dtype=np.float32)

    def _sync_stream(self, x):
        if _is_torch(x) and x.is_cuda:
            _lib.check(self._lib.mmiss_encoder_set_stream(self._h, _lib.current_stream_ptr(x.device), 0))
        else:
            _lib.check(self._lib.mmiss_encoder_set_stream(self._h, None, 1))

    def encode_image(self, pixels, out=None):
        """pixels: float32 [B,3,S,S] CLIP-normalised (numpy, or torch tensor on this GPU) -> float32 [B,proj] unit rows
        (same container kind as the input). uint8 [B,S,S,3] input takes the fused rescale+normalise path."""
        s = self.shape
        is_u8 = (pixels.dtype == np.uint8) if not _is_torch(pixels) else (str(pixels.dtype) == "torch.uint8")
        want = (s.v_image, s.v_image, 3) if is_u8 else (3, s.v_image, s.v_image)
        if tuple(pixels.shape[1:]) != want:
            raise ValueError(f"pixels must be [B,{','.join(map(str, want))}], got {tuple(pixels.shape)}")
        if not _is_torch(pixels):
            pixels = np.ascontiguousarray(pixels, dtype=np.uint8 if is_u8 else np.float32)
        elif not is_u8 and str(pixels.dtype) != "torch.float32":
            pixels = pixels.float()
        if _is_torch(pixels):
            pixels = pixels.contiguous()
        B = int(pixels.shape[0])
        out = self._out(pixels, B) if out is None else out
        with self._call_lock:  # stream hand-over + call are one unit per handle (threads: pipeline.BatchLanes)
            self._sync_stream(pixels)
            fn = self._lib.mmiss_encode_image_u8 if is_u8 else self._lib.mmiss_encode_image
            _lib.check(fn(self._h, _lib.ptr(pixels), B, _lib.ptr(out)))
        return out

    @staticmethod
    def _pack_rgb(images):
        """Sequence of uint8 [H,W,3] arrays -> (blob uint8 [sum H*W*3], offsets i64 [B], heights i32 [B], widths i32 [B])."""
        arrs = []
        for im in images:
            a = np.ascontiguousarray(im, dtype=np.uint8)
            if a.ndim != 3 or a.shape[2] != 3 or a.shape[0] < 1 or a.shape[1] < 1:
                raise ValueError(f"raw images must be uint8 [H,W,3], got {a.shape}")
            arrs.append(a)
        sizes = np.array([a.size for a in arrs], dtype=np.int64)
        offsets = np.zeros(len(arrs), dtype=np.int64)
        if len(arrs) > 1:
            offsets[1:] = np.cumsum(sizes)[:-1]
        blob = np.concatenate([a.reshape(-1) for a in arrs]) if arrs else np.zeros(0, np.uint8)
        heights = np.array([a.shape[0] for a in arrs], dtype=np.int32)
        widths = np.array([a.shape[1] for a in arrs], dtype=np.int32)
        return blob, offsets, heights, widths

    def resize_crop_rgb(self, images) -> np.ndarray:
        """Raw decoded RGB images of any size -> uint8 [B,S,S,3]: CLIPImageProcessor's resize (shortest edge, bicubic)
        + centre crop (HF:image_processing_clip.py:23-34) on the GPU, bit-identical to Pillow."""
        blob, off, hs, ws = self._pack_rgb(images)
        B, S = len(off), self.shape.v_image
        out = np.empty((B, S, S, 3), dtype=np.uint8)
        if B == 0:
            return out
        with self._call_lock:  # stream hand-over + call are one unit per handle (threads: pipeline.BatchLanes)
            self._sync_stream(blob)
            _lib.check(self._lib.mmiss_resize_crop_rgb(self._h, _lib.ptr(blob), blob.size, _lib.ptr(off), _lib.ptr(hs),
                                                       _lib.ptr(ws), B, _lib.ptr(out)))
        return out

    def encode_image_rgb(self, images, out=None) -> np.ndarray:
        """Raw decoded RGB images of any size (uint8 [H,W,3] each) -> float32 [B,proj] unit rows; resize, crop,
        rescale, normalise and the tower all run on the GPU (backend/app/utils.py:76-78 for a batch)."""
        return self.encode_image_rgb_packed(*self._pack_rgb(images), out=out)

    def encode_image_rgb_packed(self, blob, offsets, heights, widths, out=None):
        """The packed form of encode_image_rgb: blob = uint8 bytes of all images end to end (numpy, or a torch tensor
        on this GPU — then the result is a torch tensor too), image b at offsets[b], heights[b] x widths[b]."""
        off = np.ascontiguousarray(offsets, dtype=np.int64)
        hs = np.ascontiguousarray(heights, dtype=np.int32)
        ws = np.ascontiguousarray(widths, dtype=np.int32)
        B = len(off)
        if len(hs) != B or len(ws) != B:
            raise ValueError("offsets, heights and widths must have one entry per image")
        if not _is_torch(blob):
            blob = np.ascontiguousarray(blob, dtype=np.uint8).reshape(-1)
        elif str(blob.dtype) != "torch.uint8":
            raise ValueError("blob must be uint8")
        else:
            blob = blob.contiguous().view(-1)
        out = self._out(blob, B) if out is None else out
        if B == 0:
            return out
        with self._call_lock:  # stream hand-over + call are one unit per handle (threads: pipeline.BatchLanes)
            self._sync_stream(blob)
            _lib.check(self._lib.mmiss_encode_image_rgb(self._h, _lib.ptr(blob), int(blob.numel() if _is_torch(blob) else blob.size),
                                                        _lib.ptr(off), _lib.ptr(hs), _lib.ptr(ws), B, _lib.ptr(out)))
        return out

    def encode_text(self, input_ids, out=None, trim_padding: bool = True):
        """input_ids: int [B,T] (T <= context length), rows = BOS ... EOS, padding -> float32 [B,proj] unit rows.

        trim_padding: the pooled row is the first EOS and attention is causal, so columns after the LAST row's EOS
        cannot influence any output; they are dropped before the tower runs (the reference pads every query to 248
        tokens, backend/app/utils.py:88 — a 10-token prompt then costs 10/248 of the padded work). Host arrays only;
        device tensors are taken as they are (no sync to inspect them)."""
        if _is_torch(input_ids):
            import torch

            ids = input_ids.to(torch.int32).contiguous()
        else:
            ids = np.ascontiguousarray(input_ids, dtype=np.int32)
            if trim_padding and ids.ndim == 2 and ids.shape[0] > 0:
                eos = self.shape.eos_token_id
                pos = ids.argmax(axis=1) if eos == 2 else (ids == eos).argmax(axis=1)
                has = np.ones(ids.shape[0], bool) if eos == 2 else (ids == eos).any(axis=1)
                if has.all():
                    ids = np.ascontiguousarray(ids[:, : int(pos.max()) + 1])
        if ids.ndim != 2:
            raise ValueError("input_ids must be [B,T]")
        B, T = int(ids.shape[0]), int(ids.shape[1])
        out = self._out(ids, B) if out is None else out
        with self._call_lock:  # stream hand-over + call are one unit per handle (threads: pipeline.BatchLanes)
            self._sync_stream(ids)
            _lib.check(self._lib.mmiss_encode_text(self._h, _lib.ptr(ids), B, T, _lib.ptr(out)))
        return out

    # ------------------------------------------------------------------ debug
    def record_taps(self, on: bool = True):
        _lib.check(self._lib.mmiss_dbg_encoder_record_taps(self._h, 1 if on else 0))

    def set_fuse_ln(self, mode=-1):
        """-1 = automatic (folded from ~6000 rows per call, separate below; the default), 0 = separate LayerNorm kernels,
        1 = normalise during operand staging, 2 = folded into the GEMMs."""
        _lib.check(self._lib.mmiss_dbg_encoder_set_fuse_ln(self._h, int(mode)))

    def tap(self, tower: int, what: int, n: int) -> np.ndarray:
        out = np.empty(n, dtype=np.float32)
        w = C.c_int64(0)
        _lib.check(self._lib.mmiss_encoder_tap(self._h, tower, what, out.ctypes.data, n, C.byref(w)))
        return out[: w.value]

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mmiss_encoder_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
