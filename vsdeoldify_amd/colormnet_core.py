"""ColorMNet per-frame step behind the reference's class name and call shapes:

  InferenceCore   colormnet/inference/inference_core.py: step (:45-117, a frame whose first occurrence carries the ab mask) and
                  step_AnyExemplar (:119-230, a reference image of any content arrives with a frame), as colormnet_render.py:250-261
                  drives them; pad_divide_by / unpad of colormnet/util/tensor_util.py:18-51.

The network is the caller's object with the reference's four entry points (encode_key, encode_value, segment, short_term_attn;
model/network.py:52-145) -- in production vsdeoldify_amd.colormnet_net.ColorMNetNetwork, the whole network as one HIP plan; in the CPU
tests a stub or the oracle network.  This class only sequences those entry points around the memory, which is
colormnet_memory.MemoryManager on the MI355X kernels.  Sequential in time by nature (every frame reads what the previous ones
wrote): one clip per GPU, replicas only.
"""
import torch.nn.functional as F

from .colormnet_memory import MemoryManager

DIVIDE_BY = 112                       # 16 (ResNet stride) x 7: also a multiple of DINOv2's 14-pixel patches (inference_core.py:49,123)


def pad_divide_by(img, d):
    """centre-pad the last two dimensions to multiples of d; returns (padded, (left, right, top, bottom))"""
    h, w = img.shape[-2:]
    eh, ew = (-h) % d, (-w) % d
    pad = (ew // 2, ew - ew // 2, eh // 2, eh - eh // 2)
    return F.pad(img, pad), pad


def unpad(img, pad):
    if img.dim() not in (3, 4):
        raise NotImplementedError
    left, right, top, bottom = pad
    if top + bottom > 0:
        img = img[..., top:img.shape[-2] - bottom, :]
    if left + right > 0:
        img = img[..., left:img.shape[-1] - right]
    return img


class InferenceCore:
    def __init__(self, network, config, device_index=0, memory_backend=None):
        self.network, self.device_index, self._backend = network, device_index, memory_backend
        self._read_config(config)
        self.clear_memory()
        self.all_labels = None
        self.last_ti_key = self.last_ti_value = None

    def _read_config(self, config):
        self.config = config
        self.mem_every = config["mem_every"]
        self.deep_update_every = config["deep_update_every"]
        self.enable_long_term = config["enable_long_term"]
        self.deep_update_sync = self.deep_update_every < 0     # < 0: deep updates happen on memory frames

    def clear_memory(self):
        self.curr_ti = -1
        self.last_mem_ti = 0
        if not self.deep_update_sync:
            self.last_deep_update_ti = -self.deep_update_every
        self.memory = MemoryManager(self.config, device_index=self.device_index, backend=self._backend)

    def update_config(self, config):
        self._read_config(config)
        self.memory.update_config(config)

    def set_all_labels(self, all_labels):
        self.all_labels = all_labels

    # ---- the pieces both entry points share ----
    def _schedule(self, has_mask, end):
        """(memory frame?, deep update?, normal hidden update?) for the frame just entered"""
        is_mem = ((self.curr_ti - self.last_mem_ti >= self.mem_every) or has_mask) and not end
        if self.deep_update_sync:
            deep = is_mem and not end
        else:
            deep = (self.curr_ti - self.last_deep_update_ti >= self.deep_update_every) and not end
        normal = (not self.deep_update_sync or not deep) and not end
        return is_mem, deep, normal

    def _labels_differ(self, valid_labels):
        return valid_labels is None or len(self.all_labels) != len(valid_labels)

    def _read(self, key, selection, feats, normal, with_short_term=True):
        """long-range memory read (+ the short-term local attention on the last memory frame) -> decoder; returns the ab planes"""
        readout = self.memory.match_memory(key, selection).unsqueeze(0)
        if with_short_term:
            b, objs, cv, h, w = self.last_ti_value.shape
            short, _ = self.network.short_term_attn(key, self.last_ti_key, self.last_ti_value.flatten(start_dim=1, end_dim=2), None, key.shape[-2:])
            readout = readout + short.permute(1, 2, 0).view(b, objs, cv, h, w)
        hidden, _, prob = self.network.segment(feats, readout, self.memory.get_hidden(), h_out=normal, strip_bg=False)
        if normal:
            self.memory.set_hidden(hidden)
        return prob[0]

    def _memorise(self, image, key, shrinkage, selection, f16, planes, deep):
        value, hidden = self.network.encode_value(image, f16, self.memory.get_hidden(), planes.unsqueeze(0), is_deep_update=deep)
        self.memory.add_memory(key, shrinkage, value, self.all_labels, selection=selection if self.enable_long_term else None)
        self.last_mem_ti = self.curr_ti
        self.last_ti_key, self.last_ti_value = key, value
        if deep:
            self.memory.set_hidden(hidden)
            self.last_deep_update_ti = self.curr_ti

    # ---- inference_core.py:45-117: the frame itself may carry the ab planes (mask) ----
    def step(self, image, mask=None, valid_labels=None, end=False):
        """image [3,H,W], mask [2,H,W] (ab of this very frame) or None -> ab planes [2,H,W]"""
        self.curr_ti += 1
        image, self.pad = pad_divide_by(image, DIVIDE_BY)
        image = image.unsqueeze(0)
        is_mem, deep, normal = self._schedule(mask is not None, end)
        need_segment = self.curr_ti > 0 and self._labels_differ(valid_labels)
        key, shrinkage, selection, f16, f8, f4 = self.network.encode_key(image, need_ek=(self.enable_long_term or need_segment), need_sk=is_mem)
        planes = self._read(key, selection, (f16, f8, f4), normal) if need_segment else None
        if mask is not None:
            planes, _ = pad_divide_by(mask, DIVIDE_BY)
            self.memory.create_hidden_state(2, key)
        if is_mem:
            self._memorise(image, key, shrinkage, selection, f16, planes, deep)
        return unpad(planes, self.pad)

    # ---- inference_core.py:119-230: a reference image (its L planes + its ab planes) arrives with the frame ----
    def step_AnyExemplar(self, image, msk_lll=None, msk_ab=None, valid_labels=None, end=False, flag_FirstframeIsExemplar=False):
        """image [3,H,W]; msk_lll [3,H,W] = L of the reference image repeated, msk_ab [2,H,W] its ab planes (both None on ordinary frames)"""
        self.curr_ti += 1
        image, self.pad = pad_divide_by(image, DIVIDE_BY)
        image = image.unsqueeze(0)
        is_mem, deep, normal = self._schedule(msk_ab is not None, end)
        exemplar = msk_ab is not None and not flag_FirstframeIsExemplar
        need_segment = (self.curr_ti > 0 if flag_FirstframeIsExemplar else self.curr_ti >= 0) and self._labels_differ(valid_labels)
        key, shrinkage, selection, f16, f8, f4 = self.network.encode_key(image, need_ek=(self.enable_long_term or need_segment), need_sk=is_mem)
        planes = None
        if exemplar:
            # the reference image goes into the memory FIRST (its own key / value), then this frame is read against it
            need_segment, deep = True, False
            ref, _ = pad_divide_by(msk_lll, DIVIDE_BY)
            ref = ref.unsqueeze(0)
            rkey, rshrink, rsel, rf16, _, _ = self.network.encode_key(ref, need_ek=True, need_sk=is_mem)
            planes, _ = pad_divide_by(msk_ab, DIVIDE_BY)
            self.memory.create_hidden_state(2, key)
            rvalue, _ = self.network.encode_value(ref, rf16, self.memory.get_hidden(), planes.unsqueeze(0), is_deep_update=False)
            try:
                self.memory.add_memory(rkey, rshrink, rvalue, self.all_labels, selection=rsel if self.enable_long_term else None)
                self.last_mem_ti = self.curr_ti
                self.last_ti_key, self.last_ti_value = rkey, rvalue
            except Exception:                                  # inference_core.py:172-180 swallows a failing add here; so does the drop-in
                pass
        if need_segment:
            planes = self._read(key, selection, (f16, f8, f4), normal, with_short_term=not exemplar)
        if msk_ab is not None and flag_FirstframeIsExemplar:
            planes, _ = pad_divide_by(msk_ab, DIVIDE_BY)
        if is_mem:
            self._memorise(image, key, shrinkage, selection, f16, planes, deep)
        return unpad(planes, self.pad)
