"""FullSubNet+ as a drop-in ``nn.Module`` whose forward runs on libfsnp_hip.so.

Mirrors the reference's plugin surface for the inference path:

* constructor kwargs         speech_enhance/fullsubnet_plus/model/fullsubnet_plus.py:17-34
* parameter tree (strict ``load_state_dict`` of reference checkpoints,
  speech_enhance/audio_zen/inferencer/base_inferencer.py:100-107)
* ``forward(noisy_mag, noisy_real, noisy_imag) -> [B, 2, F, T]``  fullsubnet_plus.py:122-209
* attributes read by callers (``num_groups_in_drop_band``, ``look_ahead``, ...) fullsubnet_plus.py:111-117
* error behaviour: ``AssertionError`` for bad input ranks / channel counts / B == 2
  (fullsubnet_plus.py:136,141; acoustics/feature.py:263), ``NotImplementedError`` for
  unknown options (base_model.py:328, sequence_model.py:72, fullsubnet_plus.py:70).

The modules below only *hold parameters* with the reference's names; no torch op computes
anything in ``forward``.  CPU tensors are rejected: there is deliberately no CPU fallback.
"""
import ctypes
import weakref

import torch
import torch.nn as nn

from . import _lib

_TCN_DILATIONS = (1, 2, 5, 9, 1, 2, 5, 9)

# Generation counter of the process's module trees: bumped whenever ANY module registers a parameter or a submodule
# (torch's global registration hooks fire for `module.p = nn.Parameter(...)`, load_state_dict(assign=True), add_module,
# replacing a submodule, parametrize).  _HipModel._weights_key re-walks its tree when the counter has moved since it cached its
# parameter list, so a swapped Parameter OBJECT can never leave the handle on the previous checkpoint's weights.
_TREE_GENERATION = [0]


def _bump_tree_generation(*_args, **_kwargs):
    _TREE_GENERATION[0] += 1


nn.modules.module.register_module_parameter_registration_hook(_bump_tree_generation)
nn.modules.module.register_module_module_registration_hook(_bump_tree_generation)


class _StageHolder(nn.Module):
    """A parameter holder that is CALLABLE like the reference's submodule: the call runs the owning model's HIP stage kernels on the branch
    this holder belongs to (_HipModel._attach_holders sets owner, branch and the C entry point).  No torch fallback."""

    def __getstate__(self):              # (the back-reference is re-made by the owner; a weakref does not pickle)
        state = self.__dict__.copy()
        for k in ("_fsnp_owner", "_fsnp_branch", "_fsnp_entry"):
            state.pop(k, None)
        return state

    def forward(self, x):
        owner = self.__dict__.get("_fsnp_owner")
        owner = owner() if owner is not None else None
        if owner is None:
            raise RuntimeError(f"{self.__class__.__name__}: this parameter holder is not a callable stage of a fullsubnet_plus_amd model "
                               "(only the full-band attention layers and TCN stacks of FullSubNet_Plus, and the sub-band recurrent model, are)")
        return owner._stage_call(self.__dict__["_fsnp_entry"], self.__dict__["_fsnp_branch"], x)


class _TSSEParams(_StageHolder):
    """Parameter holder named like ChannelTimeSenseSELayer (attention_model.py:49-76)."""

    def __init__(self, num_channels, kersize, reduction_ratio=2):
        super().__init__()
        reduced = num_channels // reduction_ratio
        for name, k in zip(("smallConv1d", "middleConv1d", "largeConv1d"), kersize):
            setattr(self, name, nn.Sequential(nn.Conv1d(num_channels, num_channels, k, groups=num_channels)))
        self.feature_concate_fc = nn.Linear(3, 1)
        self.fc1 = nn.Linear(num_channels, reduced)
        self.fc2 = nn.Linear(reduced, num_channels)


class _SEParams(_StageHolder):
    """Parameter holder named like ChannelSELayer / ChannelCBAMLayer (attention_model.py:12-23, 302-313)."""

    def __init__(self, num_channels, reduction_ratio=2):
        super().__init__()
        self.fc1 = nn.Linear(num_channels, num_channels // reduction_ratio)
        self.fc2 = nn.Linear(num_channels // reduction_ratio, num_channels)


class _ECAParams(_StageHolder):
    """Parameter holder named like ChannelECAlayer (attention_model.py:343-347)."""

    def __init__(self, k_size=3):
        super().__init__()
        self.conv = nn.Conv1d(1, 1, kernel_size=k_size, padding=(k_size - 1) // 2, bias=False)


class _TCNBlockParams(nn.Module):
    """Parameter holder named like TCNBlock (causal_conv.py:68-94)."""

    def __init__(self, channels, hidden, kernel_size=3):
        super().__init__()
        self.conv1x1 = nn.Conv1d(channels, hidden, 1)
        self.prelu1 = nn.PReLU()
        self.norm1 = nn.GroupNorm(1, hidden, eps=1e-8)
        self.depthwise_conv = nn.Conv1d(hidden, hidden, kernel_size, groups=hidden)
        self.prelu2 = nn.PReLU()
        self.norm2 = nn.GroupNorm(1, hidden, eps=1e-8)
        self.sconv = nn.Conv1d(hidden, channels, 1)


class _FullBandParams(_StageHolder):
    """Parameter holder named like SequenceModel(sequence_model="TCN") (sequence_model.py:47-58,80-81)."""

    def __init__(self, num_freqs, hidden, output_size=None):
        super().__init__()
        self.sequence_model = nn.Sequential(*[_TCNBlockParams(num_freqs, hidden) for _ in _TCN_DILATIONS])
        self.fc_output_layer = nn.Linear(num_freqs, num_freqs if output_size is None else output_size)


class _SubBandParams(nn.Module):
    """Parameter holder named like SequenceModel(sequence_model="LSTM") (sequence_model.py:31-38,78-79)."""

    def __init__(self, input_size, hidden, output_size, kind="LSTM"):
        super().__init__()
        rnn = nn.LSTM if kind == "LSTM" else nn.GRU                      # sequence_model.py:31-46
        self.sequence_model = rnn(input_size, hidden, num_layers=2, batch_first=True)
        self.fc_output_layer = nn.Linear(hidden, output_size)

    def __getstate__(self):              # (the back-reference is re-made by the owner: _HipModel._attach_holders; a weakref does not pickle)
        state = self.__dict__.copy()
        state.pop("_fsnp_owner", None)
        return state

    def forward(self, x):
        """SequenceModel.forward (sequence_model.py:97-123) of the sub-band model as a call on the submodule, like the reference's
        `self.sb_model(sb_input)` (fullsubnet_plus.py:203): x [N, input, T] -> [N, output, T] on the fused HIP kernels of the model that
        owns this holder (fsnp_lstm2_fc).  No torch fallback: CPU tensors are refused like forward() refuses them."""
        assert x.dim() == 3, f"The shape of input is {x.shape}."          # sequence_model.py:103
        owner = self.__dict__.get("_fsnp_owner")
        owner = owner() if owner is not None else None
        if owner is None:
            raise RuntimeError("this sub-band model holder is not attached to a fullsubnet_plus_amd model")
        if not x.is_cuda:
            raise RuntimeError("fullsubnet_plus_amd runs on MI355X (HIP) only; move the model and inputs to 'cuda'. "
                               "There is deliberately no CPU fallback.")
        return owner.lstm2_fc(x)


class _HipState:
    """Owns the fsnp_handle (plain object: its finaliser must not go through nn.Module.__setattr__)."""

    def __init__(self):
        self.handle = None
        self.device = None
        self.packed_key = None

    def close(self):
        h, self.handle, self.device, self.packed_key = self.handle, None, None, None
        if h is not None:
            try:
                _lib.load().fsnp_destroy(h)
            except Exception:
                pass

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # a copied / pickled module gets its own handle lazily
    def __deepcopy__(self, memo):
        return _HipState()

    def __getstate__(self):
        return {}

    def __setstate__(self, state):
        self.__init__()


def _reference_style_init(m):
    """Behaviour of BaseModel.weight_init (base_model.py:332-397) for the module kinds on this path (through `.data`, as there)."""
    if isinstance(m, nn.Conv1d):
        nn.init.normal_(m.weight.data)
        if m.bias is not None:
            nn.init.normal_(m.bias.data)
    elif isinstance(m, nn.Linear):
        nn.init.xavier_normal_(m.weight.data)
        nn.init.normal_(m.bias.data)
    elif isinstance(m, (nn.LSTM, nn.GRU)):
        for p in m.parameters():
            (nn.init.orthogonal_ if p.dim() >= 2 else nn.init.normal_)(p.data)


def _resolve_device(device):
    """"cuda" -> the current device with its index (a handle is bound to one device)."""
    dev = torch.device(device)
    if dev.type == "cuda" and dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    return dev


class _HipModel(nn.Module):
    """Everything the two reference models share on the HIP side: the fsnp_handle, lazy strict weight packing,
    the fsnp_forward call and the test / bench hooks.  Subclasses hold the reference's parameter tree and
    provide ``_config()``."""

    def _attach_holders(self):
        """The sub-band holder is callable like the reference's submodule (`model.sb_model(x)`): it needs to know whose kernels to run."""
        sb = self._modules.get("sb_model")
        if isinstance(sb, _SubBandParams):
            sb.__dict__["_fsnp_owner"] = weakref.ref(self)
        # FullSubNet+: the three attention layers and the three full-band TCN stacks (fullsubnet_plus.py:160-165, 171-173)
        for branch, tag in enumerate(("", "_real", "_imag")):
            for name, entry in (("channel_attention" + tag, "fsnp_channel_attention"), ("fb_model" + tag, "fsnp_fullband_model")):
                mod = self._modules.get(name)
                if isinstance(mod, _StageHolder) and not isinstance(mod, (_SubBandParams, _FullBandLSTMParams)):
                    mod.__dict__.update(_fsnp_owner=weakref.ref(self), _fsnp_branch=branch, _fsnp_entry=entry)

    def __setstate__(self, state):          # copy.deepcopy / pickle: the copy's holder points at the copy
        super().__setstate__(state)
        self._attach_holders()

    def _stage_call(self, entry, branch, x):
        """One of the forward's submodules on a caller's [B, F, T] tensor (fsnp_channel_attention / fsnp_fullband_model)."""
        assert x.dim() == 3, f"The shape of input is {x.shape}."
        if not x.is_cuda:
            raise RuntimeError("fullsubnet_plus_amd runs on MI355X (HIP) only; move the model and inputs to 'cuda'. "
                               "There is deliberately no CPU fallback.")
        assert x.shape[1] == self.num_freqs, f"expected {self.num_freqs} channels, got {x.shape[1]}"
        lib = self._ensure_handle(x.device)
        x = x if x.dtype == torch.float32 else x.float()
        out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        st = (ctypes.c_int64 * 3)(*x.stride())
        stream = torch.cuda.current_stream(x.device).cuda_stream
        with torch.cuda.device(x.device):
            _lib.check(getattr(lib, entry)(self._handle, int(branch), x.data_ptr(), ctypes.byref(st), out.data_ptr(), x.shape[0], x.shape[2],
                                           ctypes.c_void_p(stream)), entry)
        return out

    def _init_hip(self):
        # "parity": B > 1 reproduces the reference's drop_band output [B,2,F//2,T] (fullsubnet_plus.py:192-196,
        # fullsubnet.py:107-110); "full": every utterance keeps all bins (== the reference run per utterance).
        self.batch_mode = "parity"
        # What happens to an asynchronous device-side failure (a column-split LSTM launch whose workgroups could not all be
        # resident - a GPU shared with another process - gives up after 2 s and flags the handle):
        #   "sync"     (default) forward() waits for its own launches, polls the flag, and - if it is set - re-runs the
        #              batch once on the one-tile-per-CU kernel (which needs no co-residency) with a warning; a result that
        #              forward() returned is therefore never silently invalid.  The reference's caller synchronises right
        #              after the call anyway (inferencer.py:151-158 moves the mask to the CPU).
        #   "deferred" nothing is waited for (throughput loops, bench.py): call check_errors() / poll_errors() before
        #              trusting results; the next call on the handle also fails loudly.
        self.error_check = "sync"
        # fsnp_watch_weights: every Nth forward starts with one fingerprint kernel over the parameters' storage (1 = every forward)
        self.weight_watch_every = 1
        # fsnp_set_verify_sample: every Nth forward of a small batch (a plan of column-split launches only) has one row tile recomputed on
        # an exchange-free kernel beside the following forwards.  None = the policy's default: 16 under error_check="sync", off under
        # "deferred" (throughput loops decide for themselves); an int = that value whatever the policy
        self.verify_sample_every = None
        self._pipeline = False
        self._hip = _HipState()

    # ------------------------------------------------------------------ BaseModel's public helpers (module protocol)
    @staticmethod
    def _stage_input(input):
        assert input.dim() == 4, f"The dim of input is {input.dim()}. It should be four dim."
        if not input.is_cuda:
            raise RuntimeError("fullsubnet_plus_amd runs on MI355X (HIP) only; move the tensor to 'cuda'. There is deliberately no CPU fallback.")
        x = input if input.dtype == torch.float32 else input.float()
        return x, (ctypes.c_int64 * 4)(*x.stride()), torch.cuda.current_stream(x.device).cuda_stream

    @staticmethod
    def unfold(input, num_neighbor):
        """BaseModel.unfold (audio_zen/model/base_model.py:15-47): [B, C, F, T] -> [B, F, C, 2 num_neighbor + 1, T], the overlapped sub-band
        units along the frequency axis (reflect padded); num_neighbor < 1: [B, F, C, 1, T].  HIP kernel (fsnp_unfold) on CUDA tensors of
        any strides.  (The forward never calls it: the recurrent kernels gather the sub-band input on the fly.)"""
        x, st, stream = _HipModel._stage_input(input)
        B, C, F, T = x.shape
        ns = 2 * int(num_neighbor) + 1 if num_neighbor >= 1 else 1
        out = torch.empty((B, F, C, ns, T), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().fsnp_unfold(x.data_ptr(), ctypes.byref(st), out.data_ptr(), B, C, F, T, int(num_neighbor), ctypes.c_void_p(stream)),
                       "fsnp_unfold")
        return out

    def norm_wrapper(self, norm_type: str):
        """BaseModel.norm_wrapper (base_model.py:318-330): -> a callable [B, C, F, T] -> [B, C, F, T] running the named normalisation as HIP
        kernels (fsnp_norm); unknown names raise NotImplementedError like the reference."""
        if norm_type not in _lib.NORM_TYPES:
            raise NotImplementedError("You must set up a type of Norm. "
                                      "e.g. offline_laplace_norm, cumulative_laplace_norm, forgetting_norm, etc.")
        code = _lib.NORM_TYPES[norm_type]

        def norm(input):
            x, st, stream = _HipModel._stage_input(input)
            B, C, F, T = x.shape
            out = torch.empty((B, C, F, T), dtype=torch.float32, device=x.device)
            with torch.cuda.device(x.device):
                _lib.check(_lib.load().fsnp_norm(code, x.data_ptr(), ctypes.byref(st), out.data_ptr(), B, C, F, T, ctypes.c_void_p(stream)), "fsnp_norm")
            return out
        norm.__name__ = norm_type
        return norm

    @property
    def norm(self):
        """`self.norm` of the reference models (fullsubnet_plus.py:115, fullsubnet.py:61): norm_wrapper(self.norm_type)."""
        return self.norm_wrapper(self.norm_type)

    def _weights_key(self):
        # (storage pointer, version) of every parameter: changes on load_state_dict, .to(), in-place updates.  The parameter
        # LIST is cached - walking the module tree (340 parameters) was 0.56 ms per forward, a quarter of a B = 1 step - and
        # rebuilt whenever Parameter OBJECTS may have been replaced: _apply (.to / .float / .cuda ...), and any parameter /
        # submodule registration anywhere in the process since the list was cached (_TREE_GENERATION: assigning a new
        # nn.Parameter to a submodule attribute, load_state_dict(assign=True), replacing a submodule, parametrize).  Deleting
        # a parameter fires no hook: every 256th call re-walks the tree regardless.
        # What NEITHER pointer nor version sees is an edit through `.data` (p.data.add_(), init.normal_(m.weight.data) - the idiom of
        # the reference's own BaseModel.weight_init, base_model.py:339-355 - EMA / weight averaging): `.data` is a fresh tensor
        # with its own version counter on the same storage.  Three nets catch it: (1) on a GPU the handle WATCHES the parameters'
        # storage (fsnp_watch_weights: one fingerprint kernel in front of every forward; forward() re-packs and re-runs under
        # error_check="sync", poll_errors() / check_errors() / the next call say so under "deferred"); (2) the periodic walk folds
        # a content fingerprint into the key (this also covers parameters that cannot be watched: on another device, not fp32);
        # (3) refresh_weights() - what weight_init() calls - forces the re-pack at once.
        cached = self.__dict__.get("_fsnp_plist")
        calls = self.__dict__.get("_fsnp_plist_calls", 0) + 1
        self.__dict__["_fsnp_plist_calls"] = calls
        if cached is None or cached[0] != _TREE_GENERATION[0] or calls % 256 == 0:
            plist = list(self.parameters())
            # (while the handle watches the parameters' storage on the device the fingerprint slot stays None: any change of the
            #  parameter set shows in the pointer tuple, re-packs, and registers the watch again)
            cached = (_TREE_GENERATION[0], plist, None if self.__dict__.get("_fsnp_watched", False) else self._content_fingerprint(plist))
            self.__dict__["_fsnp_plist"] = cached
        return (tuple([(p.data_ptr(), p._version) for p in cached[1]]), cached[2])

    @staticmethod
    def _content_fingerprint(plist):
        """(sum, sum of squares) over all parameter values in fp64: moves with any realistic in-place edit."""
        if not plist:
            return None
        with torch.no_grad():
            by_dev = {}
            for p in plist:
                by_dev.setdefault(p.device, []).append(p.detach().reshape(-1).double())
            s1 = s2 = 0.0
            for ts in by_dev.values():
                flat = torch.cat(ts)
                s1 += float(flat.sum())
                s2 += float((flat * flat).sum())
        return (s1, s2)

    def refresh_weights(self):
        """Force the next forward to re-pack the device weights from the module's parameters.  Call it after editing parameters
        through `.data` (which changes neither a storage pointer nor a version counter): `p.data.add_()`, `init.*_(m.weight.data)`,
        EMA / weight averaging.  On a GPU the handle's weight watch notices such edits by itself (see _weights_key); this is the
        explicit, immediate form - and what weight_init() uses."""
        self._hip.packed_key = None

    def weight_init(self, m):
        """BaseModel.weight_init of the reference (audio_zen/model/base_model.py:332-397), usage `model.apply(model.weight_init)`:
        Conv1d normal / normal, Linear xavier-normal / normal, LSTM / GRU orthogonal matrices and normal biases - through `.data`
        like the reference's, followed by refresh_weights()."""
        _reference_style_init(m)
        self.refresh_weights()

    def _apply(self, fn, *args, **kwargs):
        self.__dict__.pop("_fsnp_plist", None)
        return super()._apply(fn, *args, **kwargs)

    def _ensure_handle(self, device):
        lib = _lib.load()
        st = self._hip
        if st.handle is not None and st.device != device:
            st.close()
        if st.handle is None:
            with torch.cuda.device(device):
                hp = ctypes.c_void_p()
                cfg = self._config()
                _lib.check(lib.fsnp_create(ctypes.byref(cfg), ctypes.byref(hp)), "fsnp_create")
            st.handle, st.device, st.packed_key = hp, device, None
            self.__dict__.pop("_vs_applied", None)
            if self.__dict__.get("_verify_every"):
                _lib.check(lib.fsnp_set_verify(hp, int(self.__dict__["_verify_every"])), "fsnp_set_verify")
        key = self._weights_key()
        if key != st.packed_key:
            for name, tensor in self.state_dict().items():
                t = tensor.detach().to("cpu", torch.float32).contiguous()
                _lib.check(lib.fsnp_set_weight(st.handle, name.encode(), t.data_ptr(), t.numel()),
                           f"fsnp_set_weight({name})")
            with torch.cuda.device(device):
                _lib.check(lib.fsnp_commit_weights(st.handle), "fsnp_commit_weights")
            self._watch_parameters(lib, device)
            st.packed_key = self._weights_key()
        return lib

    def _watch_parameters(self, lib, device):
        """fsnp_watch_weights over the parameters' own storage - possible when every parameter is contiguous fp32 on the handle's
        device (what `.to(device)` leaves); otherwise the periodic content fingerprint of _weights_key is the net."""
        plist = self.__dict__["_fsnp_plist"][1]
        ok = bool(plist) and all(p.device == device and p.dtype == torch.float32 and p.is_contiguous() for p in plist)
        n = len(plist) if ok else 0
        ptrs = (ctypes.c_void_p * max(n, 1))(*([p.data_ptr() for p in plist] if ok else [None]))
        nums = (ctypes.c_int64 * max(n, 1))(*([p.numel() for p in plist] if ok else [0]))
        stream = torch.cuda.current_stream(device).cuda_stream
        with torch.cuda.device(device):
            _lib.check(lib.fsnp_watch_weights(self._hip.handle, ptrs, nums, n, max(1, int(self.weight_watch_every)), ctypes.c_void_p(stream)),
                       "fsnp_watch_weights")
        was = self.__dict__.get("_fsnp_watched", False)
        self.__dict__["_fsnp_watched"] = ok
        if ok != was:                       # the key's fingerprint slot follows (None while the device watches)
            self.__dict__.pop("_fsnp_plist", None)
            self._weights_key()

    @property
    def _handle(self):
        if self._hip.handle is None:
            raise RuntimeError("no HIP handle yet: run a forward on a CUDA tensor first")
        return self._hip.handle

    def _forward_impl(self, ins, batch_offset, global_batch, complex_in=False):
        """ins: 1 (FullSubNet) or 3 (FullSubNet+) tensors [B, 1, F, T], or - complex_in - ONE complex64 [B, F, T]
        tensor (the STFT itself, fsnp_forward_complex); returns the cIRM tensor."""
        noisy_mag = ins[0]
        if complex_in:
            assert noisy_mag.dim() == 3 and noisy_mag.dtype == torch.complex64
            batch_size, num_freqs, num_frames = noisy_mag.size()
        else:
            assert noisy_mag.dim() == 4
            batch_size, num_channels, num_freqs, num_frames = noisy_mag.size()
            assert num_channels == 1, f"{self.__class__.__name__} takes the mag feature as inputs."
        for t in ins[1:]:
            assert t.shape == noisy_mag.shape
        assert num_freqs == self.num_freqs, f"expected {self.num_freqs} frequency bins, got {num_freqs}"
        if not noisy_mag.is_cuda:
            raise RuntimeError("fullsubnet_plus_amd runs on MI355X (HIP) only; move the model and inputs to 'cuda'. "
                               "There is deliberately no CPU fallback.")
        device = noisy_mag.device
        gb = batch_size if global_batch is None else int(global_batch)
        parity = gb > 1 and self.batch_mode == "parity"
        if parity:
            assert gb > self.num_groups_in_drop_band, \
                f"Batch size = {gb}, num_groups = {self.num_groups_in_drop_band}. " \
                f"The batch size should larger than the num_groups."
            parity = self.num_groups_in_drop_band > 1        # drop_band with one group returns its input (feature.py:265)
        if not complex_in:
            ins = [t if t.dtype == torch.float32 else t.float() for t in ins]
        for t in ins:
            assert t.device == device
        lib = self._ensure_handle(device)
        vs = self.verify_sample_every
        vs = (16 if self.error_check == "sync" else 0) if vs is None else int(vs)
        if vs != self.__dict__.get("_vs_applied"):
            if lib.fsnp_set_verify_sample(self._handle, vs) == 0:
                self.__dict__["_vs_applied"] = vs
        out_f = num_freqs // self.num_groups_in_drop_band if parity else num_freqs
        standalone = global_batch is None
        out = torch.empty((gb if parity else batch_size, self.output_size, out_f, num_frames), dtype=torch.float32, device=device)
        if parity and not standalone:
            out.zero_()     # a shard writes only its own rows of the global tensor
        stream = torch.cuda.current_stream(device).cuda_stream
        mode = _lib.MODE_PARITY if parity else _lib.MODE_FULL
        if complex_in:
            cst = (ctypes.c_int64 * 3)(*noisy_mag.stride())
            for attempt in (0, 1):
                with torch.cuda.device(device):
                    rc = lib.fsnp_forward_complex(self._handle, torch.view_as_real(noisy_mag).data_ptr(), ctypes.byref(cst),
                                                  out.data_ptr(), batch_size, num_frames, mode, int(batch_offset), gb,
                                                  ctypes.c_void_p(stream))
                if rc != _lib.ERR_STALE_WEIGHTS or attempt:
                    break
                self._stale_weights_noticed(device)
            _lib.check(rc, "fsnp_forward_complex")
            return out
        strides = (ctypes.c_int64 * 3 * 3)()
        for i, t in enumerate(ins):
            sb, _, sf, st = t.stride()
            strides[i][0], strides[i][1], strides[i][2] = sb, sf, st
        ptrs = [t.data_ptr() for t in ins] + [None] * (3 - len(ins))
        for attempt in (0, 1):
            with torch.cuda.device(device):
                rc = lib.fsnp_forward(self._handle, ptrs[0], ptrs[1], ptrs[2],
                                      ctypes.byref(strides), out.data_ptr(), batch_size, num_frames,
                                      mode, int(batch_offset), gb, ctypes.c_void_p(stream))
            if rc != _lib.ERR_STALE_WEIGHTS or attempt:
                break
            self._stale_weights_noticed(device)       # the watch flagged an EARLIER forward: re-pack, say so, run this one
        _lib.check(rc, "fsnp_forward")
        return out

    def _stale_weights_noticed(self, device):
        import warnings
        warnings.warn("fullsubnet_plus_amd: parameters were modified in place through .data after they were packed; forwards since "
                      "that edit ran on the OLD weights (error_check='deferred' does not wait for the watch).  Re-packing now; call "
                      "model.refresh_weights() after such edits", RuntimeWarning)
        self._hip.packed_key = None
        self._ensure_handle(device)

    def _checked(self, run, device):
        """Runs `run()` (one forward) under the handle's error policy (see error_check in _init_hip)."""
        out = run()
        if self.error_check != "sync":
            return out
        if self._pipeline:
            self.flush()
        torch.cuda.current_stream(device).synchronize()
        lib = _lib.load()
        rc = lib.fsnp_poll_errors(self._handle)
        if rc == 0:
            return out
        if rc == _lib.ERR_STALE_WEIGHTS:
            # the weight watch: a parameter was edited through .data since the pack - this forward ran on the old weights.
            # Re-pack and run it again; the caller never sees the stale result (same contract as a timed-out launch).
            self._hip.packed_key = None
            out = run()
            if self._pipeline:
                self.flush()
            torch.cuda.current_stream(device).synchronize()
            _lib.check(lib.fsnp_poll_errors(self._handle), "fsnp_forward (after re-packing the weights)")
            return out
        msg = _lib.last_error()
        import warnings
        warnings.warn(f"fullsubnet_plus_amd: {msg}; re-running this batch on the one-tile-per-CU kernel", RuntimeWarning)
        prev = self.__dict__.get("_lstm_coop_mode", 1)
        _lib.check(lib.fsnp_debug_set_lstm_coop(self._handle, 0), "fsnp_debug_set_lstm_coop")
        try:
            out = run()
            if self._pipeline:
                self.flush()        # a model without a one-tile-per-CU kernel still defers its column-split chunks
            torch.cuda.current_stream(device).synchronize()
            _lib.check(lib.fsnp_poll_errors(self._handle), "fsnp_forward (retry)")
        finally:
            lib.fsnp_debug_set_lstm_coop(self._handle, prev)      # the mode the caller had set (debug_set_lstm_coop), not always 1
        return out

    def set_verify(self, every, device="cuda"):
        """Exchange verification (fsnp_set_verify): every `every`-th forward whose plan holds a column-split launch runs those
        sequences again on the one-tile-per-CU kernel (no inter-workgroup exchange) and compares on the device; a mismatch flags
        the handle (code 7: error_check="sync" then re-runs the batch on the one-tile-per-CU kernel with a RuntimeWarning,
        "deferred" reports through poll_errors() / check_errors() / the next call).  0 = off.  Also `model.verify_every = N`
        before the first forward, or FSNP_VERIFY_EVERY=N in the environment."""
        self.__dict__["_verify_every"] = int(every)
        lib = self._ensure_handle(_resolve_device(device))
        _lib.check(lib.fsnp_set_verify(self._handle, int(every)), "fsnp_set_verify")

    @property
    def verify_every(self):
        return self.__dict__.get("_verify_every", 0)

    @verify_every.setter
    def verify_every(self, every):
        self.__dict__["_verify_every"] = int(every)
        if self._hip.handle is not None:
            _lib.check(_lib.load().fsnp_set_verify(self._handle, int(every)), "fsnp_set_verify")

    def verify_count(self):
        """-> verification passes run so far (fsnp_verify_count)."""
        return int(_lib.load().fsnp_verify_count(self._handle))

    def verify_sample_stats(self):
        """-> {"samples", "skipped", "eligible_forwards"} of the sampled exchange verification (fsnp_set_verify_sample)."""
        out = (ctypes.c_int64 * 3)()
        _lib.check(_lib.load().fsnp_debug_verify_sample_stats(self._handle, ctypes.byref(out)), "fsnp_debug_verify_sample_stats")
        return {"samples": int(out[0]), "skipped": int(out[1]), "eligible_forwards": int(out[2])}

    def debug_corrupt_exchange(self, step):
        """Test hook (fsnp_debug_corrupt_exchange): the next forward's column-split launches publish one wrong h0 value at `step` - 1."""
        _lib.check(_lib.load().fsnp_debug_corrupt_exchange(self._handle, int(step)), "fsnp_debug_corrupt_exchange")

    def poll_errors(self):
        """Raise if a finished launch flagged the handle; no synchronisation (fsnp_poll_errors)."""
        _lib.check(_lib.load().fsnp_poll_errors(self._handle), "fsnp_poll_errors")

    def set_pipeline(self, enable=True, device="cuda"):
        """Pipelined serving mode (include/fsnp.h: fsnp_set_pipeline): the remainder chunks of the sub-band plan overlap
        the NEXT forward's full-band stages.  Outputs are complete only after flush()."""
        lib = self._ensure_handle(_resolve_device(device))
        _lib.check(lib.fsnp_set_pipeline(self._handle, int(bool(enable))), "fsnp_set_pipeline")
        self._pipeline = bool(enable)

    def reserve(self, max_batch, max_frames, max_samples=0, device="cuda"):
        """Grow the handle's workspace once to what any forward of <= max_batch utterances x max_frames frames needs (and the
        STFT / iSTFT area of enhance_wave for max_samples samples per utterance): a serving loop with varying clip lengths then
        never re-allocates (fsnp_reserve; stream-ordered on the current stream, no device synchronisation)."""
        dev = _resolve_device(device)
        lib = self._ensure_handle(dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        mode = _lib.MODE_PARITY if (self.batch_mode == "parity" and max_batch > 1 and self.num_groups_in_drop_band > 1) else _lib.MODE_FULL
        with torch.cuda.device(dev):
            _lib.check(lib.fsnp_reserve(self._handle, int(max_batch), int(max_frames), mode, int(max_samples), ctypes.c_void_p(stream)),
                       "fsnp_reserve")

    def flush(self):
        """Order the current stream after every deferred launch of earlier forwards (fsnp_flush)."""
        if self._hip.handle is None:
            return
        stream = torch.cuda.current_stream(self._hip.device).cuda_stream
        with torch.cuda.device(self._hip.device):
            _lib.check(_lib.load().fsnp_flush(self._handle, ctypes.c_void_p(stream)), "fsnp_flush")

    def forward_complex(self, noisy_complex, batch_offset=0, global_batch=None):
        """the forward fed with the complex64 STFT itself ([B, F, T], any strides - torch.stft's
        output is consumed in place); mag / real / imag are derived inside the HIP repack kernel instead of by the
        three torch ops of inferencer.py:143-147.  Same result as forward(|X|, X.real, X.imag)."""
        return self._checked(lambda: self._forward_impl([noisy_complex], batch_offset, global_batch, complex_in=True),
                             noisy_complex.device)

    def enhance(self, noisy_complex):
        """SURVEY.md 8(f-1) + (f-3): model forward + decompress_cIRM + complex multiply, all in HIP - lines 143-157 of
        fullsubnet_plus/inferencer/inferencer.py (`full_band_crm_mask` of fullsubnet/inferencer/inferencer.py for the
        original FullSubNet): noisy_complex [B,F,T] complex64 (torch.stft output, any strides) -> enhanced complex
        [B,F,T] ready for torch.istft.  Always keeps all bins (batch_mode "full")."""
        assert noisy_complex.dim() == 3 and noisy_complex.is_complex()
        assert self.output_size == 2, "the cIRM epilogue needs the two mask channels (decompress_cIRM, acoustics/mask.py:60-63)"
        mode, self.batch_mode = self.batch_mode, "full"
        try:
            mask = self.forward_complex(noisy_complex)
        finally:
            self.batch_mode = mode
        if self._pipeline:
            self.flush()            # the mask's deferred rows (pipelined mode) are complete on this stream only after a flush
        return self._apply_cirm(mask, noisy_complex)

    # ------------------------------------------------------------------ STFT / iSTFT in HIP
    def _wave_args(self, wav):
        assert wav.dim() == 2, "expected [B, samples]"
        if not wav.is_cuda:
            raise RuntimeError("fullsubnet_plus_amd runs on MI355X (HIP) only; there is deliberately no CPU fallback.")
        wav = wav.float()
        if wav.stride(1) != 1:
            wav = wav.contiguous()
        return wav, self._ensure_handle(wav.device), torch.cuda.current_stream(wav.device).cuda_stream

    def stft(self, wav):
        """audio_zen/acoustics/feature.py:10-31 (torch.stft, n_fft = 2 (F - 1), hop = n_fft / 2, hann, center) in HIP:
        wav [B, samples] -> complex64 [B, F, T] with torch.stft's strides."""
        wav, lib, stream = self._wave_args(wav)
        B, L = wav.shape
        T = 1 + L // (self.num_freqs - 1)
        spec = torch.empty((B, T, self.num_freqs), dtype=torch.complex64, device=wav.device)
        with torch.cuda.device(wav.device):
            _lib.check(lib.fsnp_stft(self._handle, wav.data_ptr(), wav.stride(0), torch.view_as_real(spec).data_ptr(), B, L,
                                     ctypes.c_void_p(stream)), "fsnp_stft")
        return spec.permute(0, 2, 1)

    def istft(self, spec, length):
        """feature.py:34-56 (torch.istft(..., length=length)) in HIP: complex64 [B, F, T] (any strides) -> [B, length]."""
        assert spec.dim() == 3 and spec.dtype == torch.complex64 and spec.shape[1] == self.num_freqs
        if not spec.is_cuda:
            raise RuntimeError("fullsubnet_plus_amd runs on MI355X (HIP) only; there is deliberately no CPU fallback.")
        lib = self._ensure_handle(spec.device)
        B, _, T = spec.shape
        out = torch.empty((B, int(length)), dtype=torch.float32, device=spec.device)
        st = (ctypes.c_int64 * 3)(*spec.stride())
        stream = torch.cuda.current_stream(spec.device).cuda_stream
        with torch.cuda.device(spec.device):
            _lib.check(lib.fsnp_istft(self._handle, torch.view_as_real(spec).data_ptr(), ctypes.byref(st), out.data_ptr(),
                                      out.stride(0), B, T, int(length), ctypes.c_void_p(stream)), "fsnp_istft")
        return out

    def enhance_wave(self, noisy):
        """The reference inferencer's inner loop in ONE call (fullsubnet_plus/inferencer/inferencer.py:142-158,
        `mag_complex_full_band_crm_mask`; `full_band_crm_mask` of fullsubnet/inferencer/inferencer.py for the original
        FullSubNet): noisy waveform [B, samples] -> enhanced waveform [B, samples]; STFT, model (all bins), cIRM
        decompression, complex multiply and iSTFT all run in HIP on the caller's stream."""
        assert self.output_size == 2, "the cIRM epilogue needs the two mask channels (decompress_cIRM, acoustics/mask.py:60-63)"
        wav, lib, stream = self._wave_args(noisy)
        B, L = wav.shape
        out = torch.empty((B, L), dtype=torch.float32, device=wav.device)
        def run():
            with torch.cuda.device(wav.device):
                _lib.check(lib.fsnp_enhance_wave(self._handle, wav.data_ptr(), wav.stride(0), out.data_ptr(), out.stride(0), B, L,
                                                 ctypes.c_void_p(stream)), "fsnp_enhance_wave")
            return out
        return self._checked(run, wav.device)

    def _apply_cirm(self, mask, noisy_complex):
        B, F, T = noisy_complex.shape
        out = torch.empty_strided((B, F, T), noisy_complex.stride(), dtype=torch.complex64, device=noisy_complex.device)
        xr, orr = torch.view_as_real(noisy_complex), torch.view_as_real(out)
        st = (ctypes.c_int64 * 3)(*noisy_complex.stride())
        ost = (ctypes.c_int64 * 3)(*out.stride())
        stream = torch.cuda.current_stream(noisy_complex.device).cuda_stream
        with torch.cuda.device(noisy_complex.device):
            _lib.check(_lib.load().fsnp_apply_cirm(mask.data_ptr(), xr.data_ptr(), ctypes.byref(st), orr.data_ptr(),
                                                   ctypes.byref(ost), B, F, T, ctypes.c_void_p(stream)), "fsnp_apply_cirm")
        return out

    # ------------------------------------------------------------------ test / bench helpers
    def lstm2_fc(self, x):
        """Fused sub-band LSTM + Linear alone (sequence_model.py:113-123): x [N, input, T] -> [N, 2, T]."""
        assert x.dim() == 3 and x.is_cuda
        lib = self._ensure_handle(x.device)
        n, _, steps = x.shape
        xt = x.permute(0, 2, 1).contiguous().float()
        out = torch.empty((n, self.output_size, steps), dtype=torch.float32, device=x.device)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        with torch.cuda.device(x.device):
            _lib.check(lib.fsnp_lstm2_fc(self._handle, xt.data_ptr(), out.data_ptr(), n, steps,
                                         ctypes.c_void_p(stream)), "fsnp_lstm2_fc")
        return out

    def read_stage(self, name, batch, frames):
        """Stage buffer of the last forward, time-major: [B, T', F] (gates: [B, F])."""
        lib = _lib.load()
        rows = batch if name.startswith("gate_") else batch * (frames + self.look_ahead)
        host = torch.empty((rows, self.num_freqs), dtype=torch.float32)
        _lib.check(lib.fsnp_read_stage(self._handle, name.encode(), host.data_ptr(), host.numel()), "fsnp_read_stage")
        return host if name.startswith("gate_") else host.view(batch, frames + self.look_ahead, self.num_freqs)

    def debug_set_num_cus(self, num_cus, device="cuda"):
        """Test hook: plan LSTM tiles as if the chip had `num_cus` CUs (exercises multi-round / VALU-row tiles)."""
        lib = self._ensure_handle(_resolve_device(device))
        _lib.check(lib.fsnp_debug_set_num_cus(self._handle, int(num_cus)), "fsnp_debug_set_num_cus")

    def debug_set_lstm_waves(self, waves, device="cuda"):
        """Tuning hook: waves per workgroup of the fused LSTM kernel (12 = three per SIMD, or 4)."""
        lib = self._ensure_handle(_resolve_device(device))
        _lib.check(lib.fsnp_debug_set_lstm_waves(self._handle, int(waves)), "fsnp_debug_set_lstm_waves")

    def debug_set_lstm_coop(self, mode, device="cuda"):
        """Tuning hook: 1 = use the column-split LSTM kernels for small batches (default), 0 = never, 2 = as 1 with the
        K-split kernel's serial (round-1) step schedule instead of the layer-skewed one and without the half-tile ping-pong
        kernel, 4 = as 1 + the half-tile ping-pong kernel even where FSNP_COOP_HP=0."""
        lib = self._ensure_handle(_resolve_device(device))
        _lib.check(lib.fsnp_debug_set_lstm_coop(self._handle, int(mode)), "fsnp_debug_set_lstm_coop")
        self.__dict__["_lstm_coop_mode"] = int(mode)

    def debug_set_gemm_dma(self, mode, device="cuda"):
        """Tuning hook: 1 = full-band TCN GEMMs on the LDS-DMA kernels with GroupNorm folded into the weights (default; small
        batches: the split-K kernel tcn_gemm_sk_kernel; sconv of larger problems: the 64-row kernel tcn_gemm_dma64_kernel), 2 = as 1 but
        never the small-batch kernel, 3 = the 128-row DMA kernel only, 0 = the general GEMM kernel."""
        lib = self._ensure_handle(_resolve_device(device))
        _lib.check(lib.fsnp_debug_set_gemm_dma(self._handle, int(mode)), "fsnp_debug_set_gemm_dma")

    def set_precision(self, mode, device="cuda"):
        """"fp32" (default) or "bf16_ih" (BASELINE.json configs[4]: layer-1 ih-GEMM of the sub-band LSTM in bf16; fsnp.h)."""
        assert mode in ("fp32", "bf16_ih")
        lib = self._ensure_handle(_resolve_device(device))
        _lib.check(lib.fsnp_set_precision(self._handle, {"fp32": 0, "bf16_ih": 1}[mode]), "fsnp_set_precision")

    def debug_set_chaos(self, seed, device="cuda"):
        """Test hook: drift injection for the column-split recurrent kernels (fsnp_debug_set_chaos); 0 = off."""
        lib = self._ensure_handle(_resolve_device(device))
        _lib.check(lib.fsnp_debug_set_chaos(self._handle, int(seed)), "fsnp_debug_set_chaos")

    def debug_inject_error(self):
        """Test hook: pretend an inter-workgroup wait timed out (see fsnp_debug_inject_error)."""
        _lib.check(_lib.load().fsnp_debug_inject_error(self._handle), "fsnp_debug_inject_error")

    def check_errors(self):
        """Synchronise and raise if an earlier call failed on the device (see fsnp_check_errors)."""
        _lib.check(_lib.load().fsnp_check_errors(self._handle), "fsnp_check_errors")

    def set_timing(self, enable=True):
        _lib.check(_lib.load().fsnp_set_timing(self._handle, int(bool(enable))), "fsnp_set_timing")

    def get_timing(self, reset=True):
        """-> {"lstm_ms", "fullband_ms", "forward_ms", "count"}: hipEvent sums on the forward's stream."""
        ms = (ctypes.c_double * 4)()
        cnt = (ctypes.c_int64 * 4)()
        _lib.check(_lib.load().fsnp_get_timing(self._handle, ctypes.byref(ms), ctypes.byref(cnt), int(reset)),
                   "fsnp_get_timing")
        return {"lstm_ms": ms[0], "fullband_ms": ms[1], "forward_ms": ms[2], "lstm_first_chunk_ms": ms[3], "count": int(cnt[0])}

    def launch_clock(self):
        """-> {"wall_ms", "s_memtime_ticks", "s_memtime_mhz"} of the LAST launch of the one-tile-per-CU LSTM kernel on this handle, as its
        workgroup 0 stamped them (fsnp_debug_launch_clock; synchronise first), or None if there was no such launch."""
        out = (ctypes.c_double * 7)()
        with torch.cuda.device(self._hip.device):
            if _lib.load().fsnp_debug_launch_clock(self._handle, ctypes.byref(out)) != 0:
                return None
        return {"wall_ms": out[2], "s_memtime_ticks": out[0], "s_memtime_mhz": out[3], "slowest_workgroup_ms": out[4],
                "fastest_workgroup_ms": out[5], "most_cycles_of_a_workgroup": out[6]}

    def describe_plan(self, batch, parity=False):
        """-> [{"kernel", "sequences", "tiles", "valu_rows", "precision", "workgroups", "deferred_when_pipelined"}, ...]: how the
        sub-band sequences of a `batch`-utterance forward are cut into kernel launches, the arithmetic each launch runs in under the
        current set_precision mode, and whether the pipelined serving loop sends it to the side stream (fsnp_describe_plan_ex)."""
        buf = (ctypes.c_int32 * 112)()
        n = _lib.load().fsnp_describe_plan_ex(self._handle, int(batch), int(parity), buf, 16)
        if n < 0:
            raise RuntimeError(_lib.last_error())
        names = {0: ("gru2_fc_kernel" if self.sequence_model == "GRU" else "lstm2_fc_kernel") + " (one 32-row tile per CU)",
                 1: "lstm2_coop_kernel (K split)",
                 2: "lstm2_coopn_kernel (three-way column split)", 3: "sub-band TCN",
                 4: "lstm2_fc16_kernel (one 16-row tile per CU)",
                 11: "lstm2_generic_kernel (runtime-sized fp32 FMA kernel: no tuned instantiation for these sizes)",
                 12: "lstm2_coop_hp_kernel (16 units per workgroup, gate-split waves, two half tiles per row tile in turn)",
                 13: "lstm2_coopw_kernel (a wave owns 8 / 16 units over the whole K, layer-skewed, no workgroup barrier)",
                 14: "lstm2_coop_hpw_kernel (16 units per workgroup, a wave owns 4 of them over the whole K, two half tiles per row tile in turn, no workgroup barrier)"}
        prec = {0: "f32", 1: "f32 + bf16 layer-1 ih-GEMM"}
        return [{"kernel": names[buf[7 * i]], "sequences": buf[7 * i + 1], "tiles": buf[7 * i + 2], "valu_rows": buf[7 * i + 3],
                 "precision": prec[buf[7 * i + 4]], "workgroups": buf[7 * i + 5], "deferred_when_pipelined": bool(buf[7 * i + 6])} for i in range(n)]

    def debug_set_costs(self, costs=None, workgroups_per_cu=1, device="cuda"):
        """Test hook: pin the planner's cost table (_lib.NUM_COSTS values, fsnp_get_costs order; None = built-in) and whether it
        may put two column-split workgroups on a CU (fsnp_debug_set_costs).  A shorter table prices the launch shapes it does
        not name out of every plan (19 values = no half-tile ping-pong kernel and no wave-owned column split, 21 = no wave-owned
        column split)."""
        lib = self._ensure_handle(_resolve_device(device))
        n = _lib.NUM_COSTS
        if costs is not None and len(costs) < n:
            costs = list(costs) + [1e9] * (n - len(costs))
        arr = (ctypes.c_double * n)(*costs) if costs is not None else None
        _lib.check(lib.fsnp_debug_set_costs(self._handle, arr, int(workgroups_per_cu)), "fsnp_debug_set_costs")

    @staticmethod
    def _cost_dict(v):
        return {"ksplit_us": {u: {"one_per_cu": v[2 * i], "two_per_cu": v[2 * i + 1], "one_tile": v[14 + i]} for i, u in enumerate((8, 16, 32, 64))},
                "coopn_us": {r: {"one_per_cu": v[8 + 2 * i], "two_per_cu": v[9 + 2 * i]} for i, r in enumerate((1, 2))},
                "rowtile_us": v[12], "valu_row_surcharge": v[13], "rowtile16_us": v[18],
                "halftile_pingpong_us": {"one_tile": v[19], "full_launch": v[20]},
                "coopw_us": {**{u: {"one_tile": v[23 + i], "full_launch": v[21 + i]} for i, u in enumerate((32, 64))},
                             96: {"one_tile": v[26], "full_launch": v[25]}}}

    def measure_costs(self):
        """-> the same table MEASURED on the device (fsnp_measure_costs; ~0.3 s, synchronises; the plan is not touched)."""
        buf = (ctypes.c_double * _lib.NUM_COSTS)()
        with torch.cuda.device(self._hip.device):
            _lib.check(_lib.load().fsnp_measure_costs(self._handle, ctypes.byref(buf)), "fsnp_measure_costs")
        return self._cost_dict(list(buf))

    def dump_config(self):
        """-> text: the handle's configuration and every effective FSNP_* setting (fsnp_dump_config), for bug reports."""
        lib = _lib.load()
        n = lib.fsnp_dump_config(self._handle, None, 0)
        buf = ctypes.create_string_buffer(int(n))
        lib.fsnp_dump_config(self._handle, buf, n)
        return buf.value.decode()

    def planner_costs_raw(self):
        """-> the _lib.NUM_COSTS values of fsnp_get_costs (the layout debug_set_costs takes)."""
        buf = (ctypes.c_double * _lib.NUM_COSTS)()
        _lib.check(_lib.load().fsnp_get_costs(self._handle, ctypes.byref(buf), None, None), "fsnp_get_costs")
        return list(buf)

    def planner_costs(self):
        """-> the per-step cost table (us) the sub-band planner minimises (fsnp_get_costs)."""
        buf, cal, occ = (ctypes.c_double * _lib.NUM_COSTS)(), ctypes.c_int32(), ctypes.c_int32()
        _lib.check(_lib.load().fsnp_get_costs(self._handle, ctypes.byref(buf), ctypes.byref(cal), ctypes.byref(occ)), "fsnp_get_costs")
        return {**self._cost_dict(list(buf)), "calibrated": bool(cal.value), "workgroups_per_cu": occ.value}

    def forward_flops(self, batch, frames, parity=False):
        return float(_lib.load().fsnp_forward_flops(self._handle, batch, frames, int(parity)))


class FullSubNet_Plus(_HipModel):
    def __init__(self,
                 num_freqs,
                 look_ahead,
                 sequence_model,
                 fb_num_neighbors,
                 sb_num_neighbors,
                 fb_output_activate_function,
                 sb_output_activate_function,
                 fb_model_hidden_size,
                 sb_model_hidden_size,
                 channel_attention_model="SE",
                 norm_type="offline_laplace_norm",
                 num_groups_in_drop_band=2,
                 output_size=2,
                 subband_num=1,
                 kersize=[3, 5, 10],
                 weight_init=True,
                 ):
        super().__init__()
        assert sequence_model in ("GRU", "LSTM", "TCN"), f"{self.__class__.__name__} only support GRU, LSTM and TCN."
        if sequence_model not in _lib.SEQUENCE_MODELS:
            raise NotImplementedError(f"Not implemented {sequence_model}")                 # sequence_model.py:72
        if channel_attention_model not in _lib.ATTENTION:
            raise NotImplementedError(f"Not implemented channel attention model {channel_attention_model}")
        if subband_num != 1 and channel_attention_model != "ECA":
            # fullsubnet_plus.py:47-50,155-163: the reference builds its TSSE / SE / CBAM layers for
            # num_freqs // subband_num + 1 channels and then feeds the real / imag branches num_freqs channels - its own
            # forward raises; only ECA (no per-channel parameters) runs.
            raise NotImplementedError("subband_num != 1 only works with channel_attention_model='ECA' (as in the reference)")
        if subband_num < 1:
            raise NotImplementedError("subband_num must be >= 1")
        if norm_type not in _lib.NORM_TYPES:
            raise NotImplementedError("You must set up a type of Norm. "
                                      "e.g. offline_laplace_norm, cumulative_laplace_norm, forgetting_norm, etc.")
        for act in (fb_output_activate_function, sb_output_activate_function):
            if act and act not in _lib.ACTIVATIONS:
                raise NotImplementedError(f"Not implemented activation function {act}")

        self.num_channels = num_freqs
        def make_attention():
            if channel_attention_model == "TSSE":
                return _TSSEParams(num_freqs, kersize)
            if channel_attention_model == "ECA":
                return _ECAParams()
            return _SEParams(num_freqs)                      # SE and CBAM share the parameter tree
        self.channel_attention_model = channel_attention_model
        self.channel_attention = make_attention()
        self.channel_attention_real = make_attention()
        self.channel_attention_imag = make_attention()
        # NB: the reference hard-codes the TCNBlock hidden width to 512 (causal_conv.py:68) and ignores
        # fb_model_hidden_size for the TCN full-band models (sequence_model.py:48-57).
        self.fb_model = _FullBandParams(num_freqs, 512)
        self.fb_model_real = _FullBandParams(num_freqs, 512)
        self.fb_model_imag = _FullBandParams(num_freqs, 512)
        sb_in = (sb_num_neighbors * 2 + 1) + 3 * (fb_num_neighbors * 2 + 1)
        if sequence_model == "TCN":      # sequence_model.py:47-58,80-81: TCNBlocks keep the default 512 hidden channels
            self.sb_model = _FullBandParams(sb_in, 512, output_size)
        else:
            self.sb_model = _SubBandParams(sb_in, sb_model_hidden_size, output_size, sequence_model)
        self._attach_holders()               # the submodules the forward calls are callable stages, like the reference's
        self.sequence_model = sequence_model

        self.subband_num = subband_num
        self.sb_num_neighbors = sb_num_neighbors
        self.fb_num_neighbors = fb_num_neighbors
        self.look_ahead = look_ahead
        self.norm_type = norm_type
        self.num_groups_in_drop_band = num_groups_in_drop_band
        self.output_size = output_size
        self.num_freqs = num_freqs
        self.kersize = list(kersize)
        self.fb_output_activate_function = fb_output_activate_function
        self.sb_output_activate_function = sb_output_activate_function
        self.sb_model_hidden_size = sb_model_hidden_size
        self._init_hip()
        if weight_init:
            self.apply(self.weight_init)          # fullsubnet_plus.py:119-120

    # ------------------------------------------------------------------ handle / weights
    def _config(self):
        cfg = _lib.FsnpConfig()
        cfg.num_freqs = self.num_freqs
        cfg.look_ahead = self.look_ahead
        cfg.sb_num_neighbors = self.sb_num_neighbors
        cfg.fb_num_neighbors = self.fb_num_neighbors
        cfg.tcn_hidden = 512
        cfg.num_tcn_blocks = len(_TCN_DILATIONS)
        cfg.sb_hidden = self.sb_model_hidden_size
        cfg.output_size = self.output_size
        cfg.norm_type = _lib.NORM_TYPES[self.norm_type]
        cfg.fb_act = _lib.ACTIVATIONS[self.fb_output_activate_function or None]
        cfg.sb_act = _lib.ACTIVATIONS[self.sb_output_activate_function or None]
        for i, k in enumerate(self.kersize):
            cfg.kersize[i] = int(k)
        cfg.num_groups_in_drop_band = self.num_groups_in_drop_band
        cfg.attention = _lib.ATTENTION[self.channel_attention_model]
        cfg.sequence_model = _lib.SEQUENCE_MODELS[self.sequence_model]
        cfg.subband_num = self.subband_num
        return cfg

    # ------------------------------------------------------------------ forward
    def forward(self, noisy_mag, noisy_real, noisy_imag, batch_offset=0, global_batch=None):
        """
        Shapes:
            noisy_mag / noisy_real / noisy_imag: [B, 1, F, T] fp32 CUDA tensors (any strides)
            return: [B, 2, F, T]   (B == 1 or batch_mode == "full")
                    [B, 2, F//2, T] with the reference's drop_band row order (B > 1, batch_mode == "parity")
        batch_offset / global_batch: only for sharded batches (fullsubnet_plus_amd.dist).
        """
        return self._checked(lambda: self._forward_impl([noisy_mag, noisy_real, noisy_imag], batch_offset, global_batch),
                             noisy_mag.device)


class _FullBandLSTMParams(_StageHolder):
    """Parameter holder named like SequenceModel(sequence_model="LSTM") of the original FullSubNet's full-band model
    (fullsubnet/model/fullsubnet.py:39-47; sequence_model.py:31-38,78-79)."""

    def __init__(self, num_freqs, hidden, kind="LSTM"):
        super().__init__()
        rnn = nn.LSTM if kind == "LSTM" else nn.GRU
        self.sequence_model = rnn(num_freqs, hidden, num_layers=2, batch_first=True)
        self.fc_output_layer = nn.Linear(hidden, num_freqs)


class FullSubNet(_HipModel):
    """the original FullSubNet ``Model`` (speech_enhance/fullsubnet/model/fullsubnet.py:12-118, the
    commented alternative of config/inference.toml:11,28) on the same HIP kernels: constructor kwargs
    fullsubnet.py:13-26, ``forward(noisy_mag) -> [B, 2, F, T]`` fullsubnet.py:68-118, strict state_dict."""

    def __init__(self,
                 num_freqs,
                 look_ahead,
                 sequence_model,
                 fb_num_neighbors,
                 sb_num_neighbors,
                 fb_output_activate_function,
                 sb_output_activate_function,
                 fb_model_hidden_size,
                 sb_model_hidden_size,
                 norm_type="offline_laplace_norm",
                 num_groups_in_drop_band=2,
                 weight_init=True,
                 ):
        super().__init__()
        assert sequence_model in ("GRU", "LSTM"), f"{self.__class__.__name__} only support GRU and LSTM."
        if norm_type not in _lib.NORM_TYPES:
            raise NotImplementedError("You must set up a type of Norm. "
                                      "e.g. offline_laplace_norm, cumulative_laplace_norm, forgetting_norm, etc.")
        for act in (fb_output_activate_function, sb_output_activate_function):
            if act and act not in _lib.ACTIVATIONS:
                raise NotImplementedError(f"Not implemented activation function {act}")
        self.fb_model = _FullBandLSTMParams(num_freqs, fb_model_hidden_size, sequence_model)
        self.sb_model = _SubBandParams((sb_num_neighbors * 2 + 1) + (fb_num_neighbors * 2 + 1), sb_model_hidden_size, 2,
                                       sequence_model)
        self._attach_holders()
        self.sequence_model = sequence_model

        self.sb_num_neighbors = sb_num_neighbors
        self.fb_num_neighbors = fb_num_neighbors
        self.look_ahead = look_ahead
        self.norm_type = norm_type
        self.num_groups_in_drop_band = num_groups_in_drop_band
        self.num_freqs = num_freqs
        self.output_size = 2
        self.fb_output_activate_function = fb_output_activate_function
        self.sb_output_activate_function = sb_output_activate_function
        self.fb_model_hidden_size = fb_model_hidden_size
        self.sb_model_hidden_size = sb_model_hidden_size
        self._init_hip()
        if weight_init:
            self.apply(self.weight_init)          # fullsubnet.py:65-66

    def _config(self):
        cfg = _lib.FsnpConfig()
        cfg.model = _lib.MODEL_FULLSUBNET
        cfg.num_freqs = self.num_freqs
        cfg.look_ahead = self.look_ahead
        cfg.sb_num_neighbors = self.sb_num_neighbors
        cfg.fb_num_neighbors = self.fb_num_neighbors
        cfg.tcn_hidden = self.fb_model_hidden_size       # fb_model_hidden_size (fsnp.h: FSNP_MODEL_FULLSUBNET)
        cfg.num_tcn_blocks = 0
        cfg.sb_hidden = self.sb_model_hidden_size
        cfg.output_size = 2
        cfg.norm_type = _lib.NORM_TYPES[self.norm_type]
        cfg.fb_act = _lib.ACTIVATIONS[self.fb_output_activate_function or None]
        cfg.sb_act = _lib.ACTIVATIONS[self.sb_output_activate_function or None]
        for i in range(3):
            cfg.kersize[i] = 1
        cfg.num_groups_in_drop_band = self.num_groups_in_drop_band
        cfg.attention = 0
        cfg.sequence_model = _lib.SEQUENCE_MODELS[self.sequence_model]
        return cfg

    def forward(self, noisy_mag, batch_offset=0, global_batch=None):
        """noisy_mag [B, 1, F, T] fp32 CUDA tensor (any strides) -> cIRM [B, 2, F, T] (see FullSubNet_Plus.forward
        for batch_mode and the sharding arguments)."""
        return self._checked(lambda: self._forward_impl([noisy_mag], batch_offset, global_batch), noisy_mag.device)


Model = FullSubNet_Plus  # the name BASELINE.json's north_star uses
