"""Input-side feature transforms on the device - counterpart of reference utilities/BoxTransforms.py (get_transforms,
:454-490) for the transforms the training drivers enable: ApplyLog, PadOrTrunc, TimeMask, FreqMask(fill "mean"), FreqShift,
ToTensor(unsqueeze_axis=0), Normalize(scaler).

The reference runs them per clip in NumPy inside the DataLoader workers; at > 10 k clips/s per GPU that cannot feed the step.
Here a whole batch is ONE kernel launch (sedt_box_transform: a workgroup per clip, the clip stays in LDS between the passes).
The RANDOM PARAMETERS are drawn on the host with np.random in exactly the order the reference's classes draw them
(TimeMask.randomize_parameters :380-383, FreqMask :410-413, FreqShift :437-443), so a seeded run augments every clip the same
way as the reference pipeline does."""
import numpy as np
import torch

from .. import lib as L

_AUG = np.dtype([('nframes_raw', np.int32), ('tm_t', np.int32), ('tm_t0', np.int32), ('fm_f', np.int32), ('fm_f0', np.int32),
                 ('fm_on', np.int32), ('fs_shift', np.int32), ('pad', np.int32)])


class DeviceBoxTransform(object):
    """frames: fixed number of frames (config.max_frames); scaler_mean / scaler_std: per-mel float64 vectors of the dataset
    Scaler (None: no normalisation); time_mask / freq_mask / freq_shift: enable the augmentations (their constructor
    defaults are the reference's: TimeMask(0.0, 0.1, p=0.2), FreqMask(0.03, 0.4, fill "mean", p=0.5), FreqShift(p=0.5,
    max_band=4, std=2)); apply_log=False takes inputs that are already in dB."""

    def __init__(self, frames, scaler_mean=None, scaler_std=None, time_mask=False, freq_mask=False, freq_shift=False,
                 apply_log=True, n_mels=64, device='cuda', tm=(0.0, 0.1, 0.2), fm=(0.03, 0.4, 0.5), fs=(0.5, 4, 0.0, 2.0)):
        self.frames, self.F, self.dev = frames, n_mels, torch.device(device)
        self.time_mask, self.freq_mask, self.freq_shift, self.apply_log = time_mask, freq_mask, freq_shift, apply_log
        self.tm, self.fm, self.fs = tm, fm, fs
        self.mean = self.std = None
        if scaler_mean is not None:
            self.mean = torch.as_tensor(np.asarray(scaler_mean, np.float64)).to(self.dev)
            self.std = torch.as_tensor(np.asarray(scaler_std, np.float64)).to(self.dev)

    def draw(self, nframes_raw):
        """one record of augmentation parameters for a clip, consuming np.random like the reference's transform objects"""
        r = np.zeros((), _AUG)
        r['nframes_raw'] = nframes_raw
        nf, nm = self.frames, self.F
        if self.time_mask:
            lo, hi, p = self.tm
            apply = np.random.uniform(0, 1) < p
            t = np.random.uniform(lo, hi)
            t0 = np.random.uniform(0, 1 - t)
            if apply:
                r['tm_t'], r['tm_t0'] = int(t * nf), int(t0 * nf)
        if self.freq_mask:
            lo, hi, p = self.fm
            apply = np.random.uniform(0, 1) < p
            f = np.random.uniform(lo, hi)
            f0 = np.random.uniform(0, 1 - f)
            if apply:
                r['fm_on'], r['fm_f'], r['fm_f0'] = 1, int(f * nm), int(f0 * nm)
        if self.freq_shift:
            p, max_band, mean, std = self.fs
            apply = np.random.uniform(0, 1) < p
            s = int(np.random.normal(mean, std))
            while abs(s) > max_band:
                s = int(np.random.normal(mean, std))
            if apply:
                r['fs_shift'] = s
        return r

    def __call__(self, clips, params=None, out=None, staging=None):
        """clips: list of (T_raw, n_mels) float arrays / tensors (mel amplitudes), or a (B, T_raw, n_mels) tensor already on
        the device.  params: optional structured array of _AUG records (else drawn).  Returns (B, 1, frames, n_mels) f32."""
        if torch.is_tensor(clips) and clips.is_cuda:
            amp = clips.float().contiguous()
            B, stride = amp.shape[0], amp.shape[1]
            nraw = [stride] * B
        else:
            B = len(clips)
            nraw = [int(c.shape[0]) for c in clips]
            stride = max(nraw)
            host = staging if staging is not None else torch.zeros((B, stride, self.F), dtype=torch.float32).pin_memory()
            for i, c in enumerate(clips):
                host[i, :nraw[i]].copy_(torch.as_tensor(np.asarray(c, dtype=np.float32)))
            amp = host.to(self.dev, non_blocking=True)
        if params is None:
            params = np.stack([self.draw(n) for n in nraw])
        params = np.ascontiguousarray(params)
        aug = torch.from_numpy(params.view(np.uint8).reshape(-1).copy()).to(self.dev, non_blocking=True)
        if out is None:
            out = torch.empty((B, 1, self.frames, self.F), device=self.dev, dtype=torch.float32)
        L.check(L.load().sedt_box_transform(L.p(amp), stride, L.p(aug), L.p(self.mean), L.p(self.std), B, self.frames, self.F,
                                            int(self.apply_log), 1, 0.0, L.p(out), L.stream_ptr()), 'box_transform')
        return out
