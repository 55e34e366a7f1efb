"""ColorMNetRender — drop-in for the reference's per-frame exemplar colorizer (BASELINE configs[4]):

  /root/reference/vsdeoldify/colormnet/colormnet_render.py:47-321   class ColorMNetRender (constructor, set_config, set_ref_frame,
                                                                    colorize_batch_frames, colorize_frame, get_image, the reset rule)
  called from colormnet/__init__.py:42-128 (vs_colormnet_local) and through the XML-RPC pair colormnet_server.py / colormnet_client.py
  (vs_colormnet_remote): here the call is in-process, no server thread, no PIL-bytes-over-HTTP hop.

Same constructor arguments, same methods, same state machine (frame / reference counters, the memory reset on a new reference image or when
max_memory_frames is reached, `FirstFrameIsNotExemplar`, the "no reference yet -> return the frame unchanged" rule).  The arithmetic is on the
MI355X: RGB -> normalised Lab and Lab -> RGB (csrc/zhang.hip device functions), the network (colormnet_net.ColorMNetNetwork: HIP plan over
the conv / attention / ColorMNet kernels), the memory (colormnet_memory.MemoryManager).  torch is used for device memory and tensor
bookkeeping only.  No CPU fallback: without libhavc_mi355.so or a gfx950 device the constructor raises.
Sequential in time by nature (every frame reads what the previous ones wrote) => one clip per GPU, replicas only.

image_size: the reference's HAVC entry points always pass -1 (vsdeoldify/__init__.py:1700,1712; the clip is resized beforehand by
SmartResizeColorizer, vsslib/vsresize.py:271-316); image_size >= 0 (torchvision Resize inside the transform) is refused.
"""
import collections
import os

import numpy as np

from .colormnet_core import InferenceCore

DEF_MAX_MEMORY_FRAMES = 10000              # vsslib/constants.py:64
WEIGHTS = "weights/DINOv2FeatureV6_LocalAtten_s2_154000.pth"      # colormnet_render.py:107-108


def default_config(vid_length, max_memory_frames, propagate=False):
    """the config dict of _colorize_config_init / _colorize_model_init (colormnet_render.py:95-160) for image_size = -1"""
    cfg = {"FirstFrameIsNotExemplar": not propagate, "dataset": "D16_batch"}
    cfg["max_mid_term_frames"] = min(10, vid_length)
    cfg["min_mid_term_frames"] = min(5, int(cfg["max_mid_term_frames"] / 2))
    cfg["max_long_term_elements"] = max_memory_frames
    cfg["num_prototypes"] = 128
    cfg["top_k"] = 30
    cfg["mem_every"] = min(5, cfg["max_mid_term_frames"])
    cfg["deep_update_every"] = -1
    cfg["save_scores"] = False
    cfg["size"] = -1
    cfg["disable_long_term"] = False
    cfg["enable_long_term"] = True
    span = cfg["max_mid_term_frames"] - cfg["min_mid_term_frames"]
    cfg["enable_long_term_count_usage"] = bool(span > 0 and (vid_length / span * cfg["num_prototypes"]) >= cfg["max_long_term_elements"])
    return cfg


_NETWORKS = {}


def _load_network(project_dir, state_dict, device_index):
    """one packed network per (weights, device), like the reference's singleton (`_initialized`, colormnet_render.py:68-71,92-94)"""
    from .colormnet_net import ColorMNetNetwork
    key = (id(state_dict) if state_dict is not None else os.path.join(project_dir, WEIGHTS), device_index)
    if key not in _NETWORKS:
        if state_dict is None:
            import torch
            path = key[0]
            if not os.path.isfile(path):
                raise FileNotFoundError(f"ColorMNet weights not found: {path}")
            state_dict = torch.load(path, map_location="cpu")
        _NETWORKS[key] = ColorMNetNetwork(state_dict, device_index=device_index)
    return _NETWORKS[key]


class ColorMNetRender:
    """renders one frame at a time (colormnet_render.py:47)"""

    def __init__(self, image_size=-1, vid_length=None, enable_resize=False, encode_mode=None, propagate=False, max_memory_frames=None,
                 reset_on_ref_update=True, project_dir=None, state_dict=None, device_index=0, network=None, memory_backend=None, lookahead=None):
        if image_size is not None and image_size >= 0:
            raise NotImplementedError("image_size >= 0 (resize inside the transform) is never used by HAVC (vsdeoldify/__init__.py:1700)")
        if vid_length is None:
            raise TypeError("vid_length is required (the reference computes min(DEF_MAX_MEMORY_FRAMES, vid_length) with it)")
        self.reset_on_ref_update, self.enable_resize = reset_on_ref_update, enable_resize
        self.project_dir = project_dir if project_dir is not None else os.path.dirname(os.path.realpath(__file__))
        self.encode_mode = 0 if encode_mode is None else encode_mode
        if max_memory_frames is None or max_memory_frames == 0:
            self.max_memory_frames = min(DEF_MAX_MEMORY_FRAMES, vid_length)
        else:
            self.max_memory_frames = min(DEF_MAX_MEMORY_FRAMES, max_memory_frames)
        self.vid_length, self.size = vid_length, -1
        self.total_colored_frames = self.frame_count = self.ref_count = self.ref_count_prv = 0
        self.ref_img = self.ref_img_valid = self.img = None
        self.first_mask_loaded = False
        # frames per batched key-encoder pass when the caller announces frames ahead (colorize_batch_frames / prefetch); 1 = off
        self.lookahead = int(os.environ.get("HAVC_CMN_LOOKAHEAD", "16")) if lookahead is None else int(lookahead)
        self._ahead = collections.deque()
        # set by a caller that knows its frame list (colorize_batch_frames, DeepExColorMNet.colorize_frames) right before a colorize_frame call: the NEXT call
        # is the next announced frame and brings no reference image -- the fast step may then start that frame's read early (colormnet_fast.hint_next)
        self.next_is_plain = False
        self.device_index, self._memory_backend = device_index, memory_backend          # memory_backend: CPU tests of the state machine only
        self.network = network if network is not None else _load_network(self.project_dir, state_dict, device_index)
        self.config = default_config(vid_length, self.max_memory_frames, propagate)
        self.config.update(key_dim=self.network.key_dim, value_dim=self.network.value_dim, hidden_dim=self.network.hidden_dim)
        # the frame loop on pre-sized device buffers (colormnet_fast.py) whenever the network is the HIP one; the line-by-line classes
        # (colormnet_core / colormnet_memory) otherwise: CPU tests of the state machine (memory_backend), HAVC_CMN_FAST=0
        self._fast = memory_backend is None and getattr(self.network, "fast", False) and hasattr(self.network, "fast_buffers")
        self.processor = self._new_processor()

    def _new_processor(self):
        old = getattr(self, "processor", None)
        if old is not None and hasattr(old, "drop_read_ahead"):
            old.drop_read_ahead()                                       # a read enqueued for a frame the old memory will never see
        if self._fast:
            from .colormnet_fast import FastInferenceCore
            return FastInferenceCore(self.network, self.config, device_index=self.device_index)
        return InferenceCore(self.network, self.config, device_index=self.device_index, memory_backend=self._memory_backend)

    # ---- colormnet_render.py:162-193 ----
    def set_config(self, param_name=None, param_value=None):
        self.config[param_name] = param_value
        self.processor.update_config(self.config)

    def set_ref_frame(self, frame_ref=None, frame_propagate=False):
        self.ref_img = frame_ref
        self.config["FirstFrameIsNotExemplar"] = not frame_propagate
        if frame_ref is not None:
            self.ref_img_valid = frame_ref
            self.ref_count_prv = self.ref_count if self.frame_count > 0 else 0
            self.ref_count = self.frame_count

    def colorize_batch_frames(self, frame_list=None, ref_list=None, frame_propagate=False):
        """colormnet_render.py:186-195.  The list is known up front, so the key encoder runs `lookahead` frames at a time ahead of the
        sequential step (prefetch): encode_key does not depend on the memory."""
        out, L = [], self.lookahead
        for i, (frame_i, ref_i) in enumerate(zip(frame_list, ref_list)):
            if L > 1 and i % L == 0:
                if i == 0:
                    self.prefetch(frame_list[:L])
                self.prefetch(frame_list[i + L:i + 2 * L])              # the window AFTER this one: its pass overlaps this window's frame steps
            self.set_ref_frame(ref_i, frame_propagate)
            self.next_is_plain = i + 1 < len(frame_list) and ref_list[i + 1] is None
            out.append(self.colorize_frame(i, frame_i))
        return out

    def prefetch(self, frames):
        """Look-ahead for frames that WILL be passed to colorize_frame next, in this order (the very same objects): their Lab planes and their
        keys / image features are computed now, in one batched pass of the key encoder."""
        from .device import is_device
        from .colormnet_core import pad_divide_by, DIVIDE_BY
        if not hasattr(self.network, "prefetch_keys") or len(frames) < 2 or self.lookahead <= 1:
            return
        if len({tuple(getattr(f, "shape", None) or np.asarray(f).shape) for f in frames}) != 1:
            return                                                      # frames of different sizes: no batched pass
        with self.network.on_stream():
            srcs = [f if is_device(f) else np.asarray(f) for f in frames]
            if hasattr(self.network, "prefetch_frames"):
                labs, entries = self.network.prefetch_frames(srcs, max_batch=self.lookahead)
            else:
                labs = [self.network.image_to_lab(f) for f in srcs]
                entries = self.network.prefetch_keys([pad_divide_by(lab[:1].repeat(3, 1, 1), DIVIDE_BY)[0] for lab in labs], max_batch=self.lookahead)
        for f, lab, ent in zip(frames, labs, entries):
            self._ahead.append((f, lab, ent))

    def get_frame_count(self):
        return self.frame_count

    # ---- colormnet_render.py:197-283 ----
    def colorize_frame(self, ti=None, frame_i=None):
        with self.network.on_stream():            # the step's tensor bookkeeping and the library share one HIP stream: no host sync inside a frame
            return self._colorize_frame(ti, frame_i)

    def _colorize_frame(self, ti, frame_i):
        from PIL import Image
        from .device import DeviceImage, is_device
        self.total_colored_frames += 1
        reset_1 = self.frame_count >= self.max_memory_frames          # (the reference's other trigger is < 100 MB of free device memory)
        reset_2 = self.reset_on_ref_update and self.ref_img is not None and (self.ref_count - self.ref_count_prv >= 1)
        if reset_1 or reset_2:
            self.frame_count = 0
            self.config["FirstFrameIsNotExemplar"] = True              # the reference image is the previous coloured frame
            self.processor = self._new_processor()
            ref = self.ref_img_valid
        else:
            ref = self.ref_img
            self.frame_count += 1
        if self._fast:
            return self._colorize_frame_fast(frame_i, ref)
        as_lab = lambda im: self.network.image_to_lab(im if is_device(im) else np.asarray(im))
        ahead = None                                                    # this frame went through prefetch(): its Lab planes and its key are waiting
        if self._ahead and self._ahead[0][0] is frame_i:
            ahead = self._ahead.popleft()
        elif self._ahead:                                               # the caller left the announced order: forget the look-ahead
            self._ahead.clear()
        if ahead and hasattr(self.network, "wait_prefetched"):
            self.network.wait_prefetched(ahead[2])                      # its Lab planes were computed on the look-ahead stream
        lab = ahead[1] if ahead else as_lab(frame_i)                    # [3,H,W] normalised Lab on the device (get_image :285-301)
        rgb = lab[:1].repeat(3, 1, 1)
        msk = as_lab(ref) if ref is not None else None
        if msk is not None and not self.config["FirstFrameIsNotExemplar"]:
            msk = msk[1:3]
        if not self.first_mask_loaded:
            if msk is None:
                return frame_i                                          # nothing to propagate from yet (a prefetched entry is simply dropped)
            self.first_mask_loaded = True
        labels = None
        if msk is not None:
            self.processor.set_all_labels(list(range(1, 3)))
            labels = range(1, 3)
        is_last = self.vid_length == self.total_colored_frames - 1
        if ahead:
            self.network.expect_prefetched(ahead[2])                    # the step's first encode_key call is for this frame
        if self.config["FirstFrameIsNotExemplar"]:
            if msk is None:
                prob = self.processor.step_AnyExemplar(rgb, None, None, labels, end=is_last)
            else:
                prob = self.processor.step_AnyExemplar(rgb, msk[:1].repeat(3, 1, 1), msk[1:3], labels, end=is_last)
        else:
            prob = self.processor.step(rgb, msk, labels, end=is_last)
        if is_device(frame_i):                                          # a frame that lives in HBM stays there: nothing blocks
            out = self.network.lab_to_image(lab[:1], prob, out=DeviceImage(getattr(self.network, "ctx", frame_i.ctx), frame_i.shape))
        else:
            out = Image.fromarray(self.network.lab_to_image(lab[:1], prob))
        self.img = self.ref_img_valid = out                             # save_last_image (:303-305)
        return out


    def _colorize_frame_fast(self, frame_i, ref):
        """the frame body above on the fast step (colormnet_fast.py): the frame is padded where it is converted to Lab, the step returns the padded
        ab planes, the unpad is folded into the Lab -> RGB kernel: no tensor op between the kernels of a steady-state frame"""
        from PIL import Image
        from .device import DeviceImage, is_device
        net = self.network
        ahead = None
        if self._ahead and self._ahead[0][0] is frame_i:
            ahead = self._ahead.popleft()
        elif self._ahead:
            self._ahead.clear()
        if ahead:
            net.wait_prefetched(ahead[2])
            lab, img = ahead[1], ahead[2][5]
            from .colormnet_fast import frame_pads
            pad = frame_pads(lab.shape[-2], lab.shape[-1])[0]
        elif is_device(frame_i) and getattr(net, "async_lookahead", False) and self.first_mask_loaded and ref is None and \
                (frame_i.ctx is net.ctx or getattr(frame_i, "produced_on_lookahead", False)):
            # (only frames whose buffer belongs to the network's own context -- or to the look-ahead context that will read it: the read queued on
            #  the look-ahead stream is then ordered before any reuse of the buffer, because that context's stream waits for the pass's `done` event
            #  before the step continues; a frame of a FOREIGN context could be recycled by its owner while the read is still queued: ADVICE r4)
            # a frame nobody announced, handed over in HBM by a caller that does not block on the result: its key encoder still runs on the
            # look-ahead stream (one frame per pass), so that it overlaps the memory step of the PREVIOUS frame -- consecutive colorize_frame calls
            # pipeline on the GPU (212 -> ~300 frames/s for the reference's own call shape)
            labs, entries = net.prefetch_frames([frame_i], max_batch=1)
            ahead = (frame_i, labs[0], entries[0])
            net.wait_prefetched(entries[0])
            lab, img = labs[0], entries[0][5]
            from .colormnet_fast import frame_pads
            pad = frame_pads(lab.shape[-2], lab.shape[-1])[0]
        else:
            lab, img, pad = net.frame_in(frame_i if is_device(frame_i) else np.asarray(frame_i))
        ref_img = msk_ab = None
        if ref is not None:
            ref_lab, ref_img, _ = net.frame_in(ref if is_device(ref) else np.asarray(ref))
            msk_ab = ref_lab[1:3]
        if not self.first_mask_loaded:
            if msk_ab is None:
                return frame_i
            self.first_mask_loaded = True
        labels = None
        if msk_ab is not None:
            self.processor.set_all_labels(list(range(1, 3)))
            labels = range(1, 3)
        is_last = self.vid_length == self.total_colored_frames - 1
        if ahead:
            net.expect_prefetched(ahead[2])
        plain_next, self.next_is_plain = getattr(self, "next_is_plain", False), False
        if plain_next and ahead and self._ahead and self.frame_count < self.max_memory_frames and hasattr(self.processor, "hint_next"):
            # the caller promised that the NEXT call is the next announced frame without a reference image, and it will not reset the memory: its
            # read (memory + short-term attention) may run under this frame's decoder (FastInferenceCore._read)
            self.processor.hint_next(self._ahead[0][2])
        if self.config["FirstFrameIsNotExemplar"]:
            prob = self.processor.step_AnyExemplar_padded(img, pad, ref_img, msk_ab, labels, end=is_last)
        else:
            prob = self.processor.step_padded(img, pad, msk_ab, labels, end=is_last)
        if is_device(frame_i):
            slot, self.out_slot = getattr(self, "out_slot", None), None          # DeepExColorMNet's window buffer: the frame's slot of ONE clip-shaped buffer
            if slot is not None and tuple(slot.shape) != tuple(frame_i.shape):
                slot = None
            out = net.frame_out(lab, prob, pad, out=slot if slot is not None else DeviceImage(getattr(net, "ctx", frame_i.ctx), frame_i.shape))
        else:
            out = Image.fromarray(net.frame_out(lab, prob, pad))
        self.img = self.ref_img_valid = out
        return out


DEEPEX_SIZES = {"medium": (216, 384), "fast": (144, 256), "slow": (288, 512), "slower": (360, 640)}      # get_deepex_size, deepex/__init__.py:50-83


class DeepExColorMNet:
    """The data path of HAVC_deepex(ex_model=0) (vsdeoldify/__init__.py:1665-1735) on arrays, without VapourSynth: SmartResizeColorizer
    (vsslib/vsresize.py:271-329: black borders up to the target aspect ratio, Spline64 to the DeepEx size) -> ColorMNetRender per frame, the
    reference frames arriving with the frames a scene detector flagged (colormnet/__init__.py:100-112) -> Spline64 back, borders cropped ->
    luma of the source (vs_recover_clip_luma).  Scene detection itself, the HAVC-generated reference clip, ref_merge and the dark / smooth /
    colormap tweaks of the reference frames stay in VapourSynth: pass the reference images.  Spline64 = the library's (zimg is outside the
    parity contract)."""

    def __init__(self, vid_length, render_speed="medium", enable_resize=False, render_vivid=True, max_memory_frames=0, frame_propagate=False,
                 project_dir=None, state_dict=None, device_index=0, network=None):
        if render_speed.lower() not in DEEPEX_SIZES:
            raise ValueError("HAVC_deepex: unknown render_speed ->" + render_speed)
        scale = 2 if enable_resize else 1
        self.th, self.tw = (v * scale for v in DEEPEX_SIZES[render_speed.lower()])
        if max_memory_frames > 0:
            render_vivid = False                                          # __init__.py:1692-1693
        self.propagate = frame_propagate
        self.render = ColorMNetRender(image_size=-1, vid_length=vid_length, enable_resize=enable_resize, encode_mode=1, max_memory_frames=max_memory_frames,
                                      reset_on_ref_update=render_vivid, project_dir=project_dir, state_dict=state_dict, device_index=device_index,
                                      network=network)
        self.ctx = self.render.network.ctx
        self.t = 0
        self._announced = {}                                                 # id(frame) -> (frame, squashed frame + borders): look-ahead of colorize_frames

    def _borders(self, h, w):
        """SmartResizeColorizer.get_resized_clip: (pad_h, pad_w) of black borders that bring the clip to the target aspect ratio"""
        rt, rc = round(self.tw / self.th, 2), round(w / h, 2)
        if rc < rt:
            return 0, int(round((round(h * rt, 0) - w) / 2, 0))
        if rc > rt:
            return int(round((round(w / rt, 0) - h) / 2, 0)), 0
        return 0, 0

    def _squash(self, img, ctx=None):
        from .havc import spline64
        h, w = img.shape[-3], img.shape[-2]                                  # (a frame, or a window of frames [n, h, w, 3] without borders)
        ph, pw = self._borders(h, w)
        if ph or pw:
            from .device import is_device
            a = img.numpy() if is_device(img) else np.asarray(img)
            img = np.pad(a, ((ph, ph), (pw, pw), (0, 0)))                  # std.AddBorders: black
        return spline64(ctx or self.ctx, img, self.tw, self.th), (ph, pw)

    def _small(self, frame, ctx=None):
        """the frame as ColorMNetRender gets it (DeviceImage, or a PIL image) + the borders that were added"""
        from PIL import Image
        from .device import is_device
        small, pads = self._squash(frame, ctx)
        return (small if is_device(small) else Image.fromarray(small)), pads

    def colorize_frame(self, frame, ref=None, _small=None, _slot=None, _next_plain=False):
        """frame: u8 [h, w, 3] (ndarray or DeviceImage); ref: the reference image for THIS frame (same size as the clip) or None.
        _slot (colorize_frames): a frame of a window-sized low-resolution buffer -- the coloured small frame is written there and RETURNED without the
        Spline64 pass back to the clip size, which the caller then runs once for the whole window.  _next_plain (colorize_frames): the next call is the
        next announced frame and carries no reference image (ColorMNetRender._colorize_frame_fast may start its read early)."""
        from .device import is_device
        from .havc import spline64
        h, w = frame.shape[:2]
        if _small is None and is_device(frame) and frame.complete and ref is None and self.render.first_mask_loaded and frame.ctx is self.render.network.ctx:
            # a resident frame nobody announced: squashed on the look-ahead context, where its key encoder will run -- nothing of this frame's
            # preparation waits for the memory step of the previous one (ColorMNetRender._colorize_frame_fast)
            net = self.render.network
            la_ctx = net.lookahead_context() if hasattr(net, "lookahead_context") and getattr(net, "async_lookahead", False) else None
            if la_ctx is not None:
                _small = self._small(frame, la_ctx)
                _small[0].produced_on_lookahead = True
        small, (ph, pw) = _small if _small is not None else self._small(frame)
        if ref is not None:
            rs, _ = self._squash(ref)
            ref = rs if is_device(rs) else np.asarray(rs)
        self.render.set_ref_frame(ref, self.propagate)
        self.render.next_is_plain = bool(_next_plain)
        if _slot is not None and is_device(small) and not (ph or pw) and self.render.first_mask_loaded:
            self.render.out_slot = _slot
            col = self.render.colorize_frame(self.t, small)
            self.render.out_slot = None                                      # (a path that did not consume the slot must not find it on a later frame)
            self.t += 1
            if col is _slot:
                return col
        else:
            col = self.render.colorize_frame(self.t, small)
            self.t += 1
        col = col if is_device(col) else np.asarray(col)
        if ph or pw:                                                         # restore_clip_size: Spline64 to the bordered size, crop, then the luma
            up = spline64(self.ctx, col, w + 2 * pw, h + 2 * ph)
            up = (up.numpy() if is_device(up) else up)[ph:ph + h, pw:pw + w]
            from . import imfilters as F
            return F.chroma_post_process_np(self.ctx, np.ascontiguousarray(up), frame.numpy() if is_device(frame) else np.asarray(frame))
        return spline64(self.ctx, col, w, h, luma_from=frame)

    def _announce(self, frames):
        """squash frames that will be coloured next and hand them to the render's look-ahead (once per frame object)"""
        new = [f for f in frames if id(f) not in self._announced]
        if len(new) < 2 or self.render.lookahead <= 1:
            return
        net = self.render.network                                            # frames announced ahead are squashed on the look-ahead context's stream
        la_ctx = net.lookahead_context() if hasattr(net, "lookahead_context") else None
        if la_ctx is not None:
            # A device frame may still be being WRITTEN on its own context's stream (the output of another model handed straight to deepex):
            # the look-ahead stream must not read it unordered (device.py's rule: a buffer crosses contexts behind its producer).  Frames of the
            # network's own context are ordered ON THE GPU (the look-ahead stream waits for that stream's tail: no host stall, the memory step
            # keeps running); any other producer context is drained once per window.
            from .device import is_device
            for pctx in {id(f.ctx): f.ctx for f in new if is_device(f) and f.ctx is not la_ctx and not f.complete}.values():
                if pctx is net.ctx and hasattr(net, "lookahead_wait_for_main"):
                    net.lookahead_wait_for_main()
                else:
                    pctx.synchronize()
        if la_ctx is not None and len(new) > 1 and self._window_is_contiguous(new):
            # a resident clip: ONE Spline64 squash for the whole window (two launches instead of two per frame), on the look-ahead stream
            from .device import DeviceImage
            h, w = new[0].shape[:2]
            src = DeviceImage(new[0].ctx, (len(new), h, w, 3), ptr=new[0].ptr, owner=new[0])
            src._keep = new
            batch, pads = self._squash(src, la_ctx)
            smalls = [(batch.frame(j), pads) for j in range(len(new))]
        else:
            smalls = [self._small(f, la_ctx) for f in new]
        if la_ctx is not None:
            from .device import is_device as _isdev
            for sm, _ in smalls:
                if _isdev(sm):
                    sm.produced_on_lookahead = True                          # written on the very stream that will read it: no cross-stream wait
        self.render.prefetch([s for s, _ in smalls])
        for f, sm in zip(new, smalls):
            self._announced[id(f)] = (f, sm)                                 # (holding f keeps its id unique)

    def colorize_frames(self, frames, refs, upcoming=()):
        """frames: a sequence of u8 [h, w, 3] frames (ndarrays or DeviceImages); refs: {index: reference image} -> list of coloured frames.
        The frames are known up front: they are squashed `lookahead` at a time and announced to the render (ColorMNetRender.prefetch) ONE WINDOW
        AHEAD, so that the key-encoder pass of the next window runs (on its own stream) while the memory step walks this window frame by frame.
        upcoming: the frame objects the NEXT call will start with (a streaming caller knows them): announced during the last window of this call."""
        from .device import DeviceImage
        frames, out, L = list(frames), [], max(1, self.render.lookahead)
        if self._announced:                                                  # frames announced as `upcoming` that the caller did not come back with:
            live = {id(f) for f in frames}                                   # their squashed copies and look-ahead entries are dropped, not kept forever
            for k in [k for k in self._announced if k not in live]:          # (the render's own FIFO forgets an abandoned order by itself)
                del self._announced[k]
        for i0 in range(0, len(frames), L):
            if i0 == 0:
                self._announce(frames[:L])
            self._announce(frames[i0 + L:i0 + 2 * L] or list(upcoming)[:L])
            win = frames[i0:i0 + L]
            # Round 5: the Spline64 pass back to the clip size + the luma of the source run ONCE per window (two launches for 16 frames instead of two per
            # frame: 114 -> ~40 us of a ~1.1 ms frame) when the window's frames sit back to back in one device buffer (a resident clip) and no frame
            # of the window carries a reference image; the small coloured frames go straight into the slots of one window buffer.
            batched = self._window_is_contiguous(win) and not any((i0 + j) in refs for j in range(len(win))) and self.render.first_mask_loaded and len(win) > 1
            slots = DeviceImage(self.ctx, (len(win), self.th, self.tw, 3)) if batched else None
            cols = []
            for j, f in enumerate(win):
                ent = self._announced.pop(id(f), None)
                nxt = i0 + j + 1
                cols.append(self.colorize_frame(f, refs.get(i0 + j), _small=ent[1] if ent else None, _slot=slots.frame(j) if batched else None,
                                                _next_plain=nxt < len(frames) and nxt not in refs))
            if batched and all(c is not None and getattr(c, "_owner", None) is slots for c in cols):
                from .havc import spline64
                h, w = win[0].shape[:2]
                src = DeviceImage(win[0].ctx, (len(win), h, w, 3), ptr=win[0].ptr, owner=win[0])   # the window of the clip as ONE operand (luma source)
                src._keep = win
                up = spline64(self.ctx, slots, w, h, luma_from=src)
                out.extend(up.frame(j) for j in range(len(win)))
            else:                                                            # (a frame fell back to the per-frame path: finish the others one by one)
                from .havc import spline64
                for f, c in zip(win, cols):
                    out.append(spline64(self.ctx, c, f.shape[1], f.shape[0], luma_from=f) if (batched and getattr(c, "_owner", None) is slots) else c)
        return out

    def _window_is_contiguous(self, win):
        """device frames of one size and one context, each starting where the previous one ends, no borders to add"""
        from .device import is_device
        if not win or not all(is_device(f) and f.ndim == 3 for f in win):
            return False
        f0 = win[0]
        if self._borders(f0.shape[0], f0.shape[1]) != (0, 0) or f0.ctx is not self.ctx:
            return False
        base = f0.ptr.value if hasattr(f0.ptr, "value") else int(f0.ptr)
        for j, f in enumerate(win):
            p = f.ptr.value if hasattr(f.ptr, "value") else int(f.ptr)
            if tuple(f.shape) != tuple(f0.shape) or f.ctx is not f0.ctx or p != base + j * f0.nbytes:
                return False
        return True

    def colorize_clip(self, clip, refs):
        """clip: u8 [n, h, w, 3]; refs: {frame index: reference image}; -> u8 [n, h, w, 3]"""
        return np.stack([np.asarray(o) for o in self.colorize_frames(clip, refs)])
