"""torch.optim.Adam with the update of every tensor in one HIP launch.

Drop-in for the two optimizers of the search loop (reference mmimdb_darts_searchable.py:28-33:
`Adam(central_params, lr=eta_max, weight_decay=wd)` and `Adam(arch_parameters, betas=(0.5, 0.999),
weight_decay=arch_wd)`): same constructor, same `param_groups`, same `state` layout (`step`,
`exp_avg`, `exp_avg_sq` per parameter, so `state_dict()` / `load_state_dict()` interoperate with
torch.optim.Adam checkpoints), same arithmetic operation by operation.  What changes is the
execution: the step-dependent scalars are computed on the host in double (as torch does), written
with the tensor descriptors into ONE pinned staging buffer, copied to the device with one async
H2D copy, and one `bmnas_adam_multi` launch updates every tensor (torch's foreach path: ~10
launches per parameter group).  Because all per-step values travel through the staging buffer,
a step captured in a hipGraph replays correctly: `prepare_replay()` before each replay.
"""
import math

import numpy as np
import torch

from . import lib

# bmnas_adam_tensor_t (include/bmnas_hip.h)
_DESC = np.dtype([('param', '<u8'), ('grad', '<u8'), ('exp_avg', '<u8'), ('exp_avg_sq', '<u8'),
                  ('numel', '<i8'), ('hyp_row', '<i4'), ('reserved', '<i4')])
assert _DESC.itemsize == 48


class Adam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if amsgrad:
            raise lib.BmnasError('bmnas.optim.Adam: amsgrad is not on the reference path')
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False)
        self._plan = None
        self._gen = 0            # bumped whenever the state tensors are replaced (load_state_dict)
        self._touch = 0          # bumped by everything that may re-point .grad or move the step counts outside a replay

    # ------------------------------------------------------------------ plan
    def _active(self):
        return [(p, gi) for gi, g in enumerate(self.param_groups) for p in g['params'] if p.grad is not None]

    def _init_state(self, p):
        st = self.state[p]
        if len(st) == 0:
            st['step'] = torch.tensor(0.0, dtype=torch.float32)
            st['exp_avg'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
        return st

    def _flush_counts(self):
        """Write the plan's step counts back into the per-parameter `step` tensors."""
        pl = self._plan
        if pl is not None:
            for (p, _), r in zip(pl['active'], pl['row_of']):
                self.state[p]['step'].fill_(pl['count'][r])

    def _build(self, active, slots=1):
        """slots > 1: a captured graph that holds `slots` consecutive optimizer steps (bmnas.graph.GraphedTrainStep(k=...)):
        every step has its own scalar rows and its own descriptor table (each step's backward leaves its gradients in
        tensors of its own), all in the ONE staging buffer."""
        self._flush_counts()
        dev = active[0][0].device
        E = lib.adam_chunk_elems()
        rows, row_idx, row_of, chunks = [], {}, [], []
        for i, (p, gi) in enumerate(active):
            if not p.is_cuda:
                raise lib.BmnasError('bmnas.optim.Adam needs parameters on the GPU (HIP kernel, no CPU path)')
            st = self._init_state(p)
            for name, t in (('parameter', p), ('exp_avg', st['exp_avg']), ('exp_avg_sq', st['exp_avg_sq'])):
                if t.device != dev or t.dtype != torch.float32 or not t.is_contiguous():
                    raise lib.BmnasError(f'bmnas.optim.Adam: {name} must be contiguous fp32 on {dev}')
            # parameters of one group with the same step count share a row of scalars
            key = (gi, float(st['step']))
            if key not in row_idx:
                row_idx[key] = len(rows)
                rows.append(key)
            row_of.append(row_idx[key])
            chunks += [(i, c) for c in range((p.numel() + E - 1) // E)]
        # (scalar slots packed back to back — 32 bytes per row —: the whole region travels by value in poke mode)
        hyp_slot = len(rows) * 32
        hyp_bytes = (slots * hyp_slot + 63) // 64 * 64
        tab_slot = len(active) * _DESC.itemsize
        nbytes = hyp_bytes + slots * tab_slot
        pin = torch.zeros(nbytes, dtype=torch.uint8).pin_memory()
        host = pin.numpy()
        tabs = [host[hyp_bytes + i * tab_slot:hyp_bytes + (i + 1) * tab_slot].view(_DESC) for i in range(slots)]
        for tab in tabs:
            tab['exp_avg'] = [self.state[p]['exp_avg'].data_ptr() for p, _ in active]
            tab['exp_avg_sq'] = [self.state[p]['exp_avg_sq'].data_ptr() for p, _ in active]
            tab['numel'] = [p.numel() for p, _ in active]
            tab['hyp_row'] = row_of
        devbuf = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        hyps = [host[i * hyp_slot:(i + 1) * hyp_slot].view(np.float32).reshape(len(rows), 8) for i in range(slots)]
        self._plan = dict(ids=tuple(id(p) for p, _ in active), active=active, row_of=row_of, gen=self._gen,
                          groups=[gi for gi, _ in rows], count=[t for _, t in rows],
                          pin=pin, hyp=hyps[0], hyps=hyps, hyp_all=host[:slots * hyp_slot], tab=tabs[0], tabs=tabs,
                          dev=devbuf, dev_hyp=devbuf[:hyp_bytes],
                          dev_hyps=[devbuf[i * hyp_slot:(i + 1) * hyp_slot] for i in range(slots)],
                          dev_tab=devbuf[hyp_bytes:hyp_bytes + tab_slot],
                          dev_tabs=[devbuf[hyp_bytes + i * tab_slot:hyp_bytes + (i + 1) * tab_slot]
                                    for i in range(slots)],
                          chunks=torch.tensor(chunks, dtype=torch.int32).reshape(-1, 2).to(dev),
                          n_chunks=len(chunks), slots=slots, cap_slot=0)

    def _stage(self, active):
        """Advance the step counts; write this step's scalars and pointers into the staging buffer."""
        self.prepare_replay()
        self._write_ptrs(active)

    def _write_ptrs(self, active, slot=0):
        grads = [p.grad for p, _ in active]
        for gr, (p, _) in zip(grads, active):
            if gr.dtype != torch.float32 or not gr.is_contiguous() or gr.device != p.device:
                raise lib.BmnasError('bmnas.optim.Adam: gradients must be contiguous fp32 on the parameter device')
        tab = self._plan['tabs'][slot]
        tab['param'] = [p.data_ptr() for p, _ in active]
        tab['grad'] = [gr.data_ptr() for gr in grads]

    def _launch(self, slot=0):
        pl = self._plan
        if not pl.get('poke') and slot == 0:
            pl['dev'].copy_(pl['pin'], non_blocking=True)       # (one copy node serves every slot of the graph)
        lib.adam_multi(pl['dev_tabs'][slot], pl['chunks'], pl['n_chunks'], pl['dev_hyps'][slot])

    def wait_staging(self):
        """Block until the last launch has consumed the pinned staging buffer (the host may run
        ahead of the GPU; rewriting the buffer earlier would change a step still in flight)."""
        if self._plan is not None and self._plan.get('event') is not None:
            self._plan['event'].synchronize()

    def mark_launched(self, force=False):
        """Record, on the current stream, that everything queued so far (a step() or a graph
        replay that contains one) has read the staging buffer.  (Poke-mode plans: their replays do not read it —
        nothing to record, and nothing for the next call to wait for: the host may run ahead of the GPU.)"""
        pl = self._plan
        if pl.get('poke') and not force:
            return
        if pl.get('event') is None:
            pl['event'] = torch.cuda.Event()
        pl['event'].record()

    # ------------------------------------------------------------------ torch.optim API
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        active = self._active()
        if not active:
            return loss
        self._touch += 1
        if torch.cuda.is_current_stream_capturing():
            if self._plan is None or self._plan['ids'] != tuple(id(p) for p, _ in active):
                raise lib.BmnasError('bmnas.optim.Adam: call capture_safe() before capturing step() in a graph')
            slot = self._plan['cap_slot']
            if slot >= self._plan['slots']:
                raise lib.BmnasError('bmnas.optim.Adam: more captured steps than capture_safe(slots=...) planned')
            self._write_ptrs(active, slot)      # the capture's static gradient tensors (of THIS step of the graph)
            self._plan['active'] = active
            self._plan.setdefault('slot_grads', {})[slot] = [p.grad for p, _ in active]
            self._launch(slot)                  # scalars are refreshed by prepare_replay()
            self._plan['cap_slot'] = slot + 1
            return loss
        ids = tuple(id(p) for p, _ in active)
        if self._plan is None or self._plan.get('captured') or self._plan['ids'] != ids:
            # never stage an eager step through a captured plan: its graph re-reads those buffers
            eager = getattr(self, '_eager_plan', None)
            if eager is not None and eager['ids'] == ids:
                self._switch(eager)
            else:
                self._build(active)
                self._eager_plan = self._plan
        if self._plan['gen'] != self._gen:
            self._switch(self._plan)
        self.wait_staging()
        self._stage(active)
        self._launch()
        self.mark_launched()
        return loss

    def zero_grad(self, set_to_none=True):
        self._touch += 1         # .grad re-pointed / dropped: a captured plan re-attaches its static tensors next time
        return super().zero_grad(set_to_none)

    def state_dict(self):
        self._flush_counts()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        for st in self.state.values():
            if torch.is_tensor(st.get('step')) and st['step'].is_cuda:
                st['step'] = st['step'].cpu()
        # the moment tensors are new objects: eager steps rebuild their plan, a captured plan
        # (held by its GraphedTrainStep) re-reads pointers and counts in activate()
        self._plan = None
        self._eager_plan = None
        self._gen += 1

    # ------------------------------------------------------------------ hipGraph support
    def capture_safe(self, poke=False, slots=1):
        """Build a plan OF ITS OWN from the gradients that exist NOW (static tensors of the step
        being captured) so that step() inside `torch.cuda.graph` issues only stream work: one
        pinned H2D copy and one launch.  The captured graph keeps reading this plan's staging
        buffers, so the plan is never reused for eager steps: take it with `captured_plan()` after
        the capture and call `activate(plan)` + `prepare_replay()` before every replay."""
        active = self._active()
        self.wait_staging()
        self._build(active, slots)
        self._write_ptrs(active)
        self._plan['captured'] = True
        # "poke" mode (round 5): the captured step holds NO H2D copy node.  The descriptor table is uploaded once after the
        # capture (and again whenever activate() / load_state_dict changed it); the per-step scalars — 32 bytes per row —
        # travel by value in the launch that copies the batch into the step's static tensors (bmnas_copy_batch's blob,
        # replay_blob()).  A captured copy node cost 4.7 us per optimizer step.
        self._plan['poke'] = bool(poke) and self._plan['hyp_all'].nbytes <= lib.copy_blob_max()
        if slots > 1 and not self._plan['poke']:
            raise lib.BmnasError(f'bmnas.optim.Adam: {slots} captured steps x {len(self._plan["groups"])} scalar rows do not '
                                 'fit the by-value blob of the batch-copy launch')

    def captured_plan(self):
        """The plan a capture has just baked into a graph, with the pointer table as captured
        (parameters, static gradient tensors) snapshotted."""
        pl = self._plan
        if pl is None or not pl.get('captured'):
            raise lib.BmnasError('bmnas.optim.Adam: no captured plan (capture_safe() + a captured step() first)')
        if pl['cap_slot'] != pl['slots']:
            raise lib.BmnasError(f'bmnas.optim.Adam: the capture holds {pl["cap_slot"]} step(s), the plan was built for '
                                 f'{pl["slots"]}')
        pl['snap_param'] = [t['param'].copy() for t in pl['tabs']]
        pl['snap_grad'] = [t['grad'].copy() for t in pl['tabs']]
        # keeps every slot's static gradient tensors alive; .grad is re-attached to the LAST step's (what a sequence of
        # eager steps leaves behind)
        pl['static_grads'] = pl['slot_grads'][pl['slots'] - 1]
        pl['tab_dirty'] = True                                       # poke mode: upload before the first replay
        return pl

    def replay_blob(self):
        """Poke mode, after prepare_replay(): -> (device tensor, host bytes) of this replay's scalar rows for
        bmnas_copy_batch's blob; uploads the descriptor table first when it changed (eagerly, on the current stream, i.e.
        in front of the replay).  None: the captured step carries its own H2D copy node."""
        pl = self._plan
        if not pl.get('poke'):
            return None
        if pl.get('tab_dirty'):
            self.wait_staging()
            pl['dev'].copy_(pl['pin'], non_blocking=True)
            self.mark_launched(force=True)       # the pinned buffer must not be rewritten before this copy has read it
            self.wait_staging()                  # (rare: once after a capture / load_state_dict)
            pl['tab_dirty'] = False
        h = pl['hyp_all']
        return pl['dev_hyp'][:h.nbytes], h.tobytes()

    def _switch(self, plan):
        """Make `plan` current: step counts travel through the per-parameter `step` tensors, the
        moment pointers are re-read when load_state_dict() replaced the state tensors."""
        if self._plan is not plan:
            self._flush_counts()
            counts = {}
            for (p, _), r in zip(plan['active'], plan['row_of']):
                c = float(self._init_state(p)['step'])
                if counts.setdefault(r, c) != c:
                    raise lib.BmnasError('bmnas.optim.Adam: parameters that shared a step count when this plan was '
                                         'built have diverged (an eager step on a subset?); re-capture the graph')
            for r, c in counts.items():
                plan['count'][r] = c
            self._plan = plan
        if not plan.get('poke'):
            self.wait_staging()              # the plan's last launch has consumed its pinned buffer
        if plan['gen'] != self._gen:
            for tab in plan['tabs']:
                tab['exp_avg'] = [self._init_state(p)['exp_avg'].data_ptr() for p, _ in plan['active']]
                tab['exp_avg_sq'] = [self.state[p]['exp_avg_sq'].data_ptr() for p, _ in plan['active']]
            plan['gen'] = self._gen

    def activate(self, plan):
        """Make `plan` (from captured_plan()) the current one before its graph is replayed.  Eager
        steps in between — a ragged last batch, the Architect's eager path — run on a plan of their
        own, so the staging buffers the graph reads are intact; but they advanced the step counts
        and re-pointed `.grad`.  The counts are carried over, the capture-time parameter / gradient
        pointers are restored and `.grad` is re-attached to the static tensors the graph writes."""
        # fast path (every replay of a training loop): this plan is current, no eager step / zero_grad / load_state_dict
        # happened since its last activation — nothing to restore
        if self._plan is plan and plan['gen'] == self._gen and plan.get('stamp') == self._touch:
            if not plan.get('poke'):
                # the captured step carries an H2D copy node that reads this plan's pinned staging buffer (capture_safe()
                # without poke, or more scalar rows than the blob holds): prepare_replay() is about to rewrite that
                # buffer, so the previous replay's copy must have read it first
                self.wait_staging()
            return
        gen = plan['gen']
        self._switch(plan)
        if plan['gen'] != gen:
            plan['tab_dirty'] = True             # load_state_dict replaced the moment tensors: new pointers in the table
        for tab, sp, sg in zip(plan['tabs'], plan['snap_param'], plan['snap_grad']):
            if plan.get('poke') and ((tab['param'] != sp).any() or (tab['grad'] != sg).any()):
                plan['tab_dirty'] = True
            tab['param'] = sp
            tab['grad'] = sg
        for (p, _), gr in zip(plan['active'], plan['static_grads']):
            p.grad = gr
        plan['stamp'] = self._touch

    def prepare_replay(self, slot=0):
        """Advance the step count and publish the current learning rates for the next replay
        (call wait_staging() first and mark_launched() after the replay; GraphedTrainStep does).
        slot: which of the graph's consecutive steps these scalars are for — a graph of k steps takes k calls, slot 0
        first, each with the learning rates its step runs with (the scheduler moves them between two calls)."""
        pl = self._plan
        h = pl['hyps'][slot]
        for r, gi in enumerate(pl['groups']):
            g = self.param_groups[gi]
            pl['count'][r] += 1.0
            t = pl['count'][r]
            b1, b2 = g['betas']
            h[r] = (-(g['lr'] / (1 - b1 ** t)), math.sqrt(1 - b2 ** t), b1, b2, g['eps'], g['weight_decay'],
                    1 - b1, 1 - b2)

    def staged_lrs(self, slot=0):
        """The learning rate of every scalar row of `slot`, decoded from what prepare_replay() STAGED there (the fp32 value
        the Adam launch of that step reads) — for loops that report the rate a batch's step ran with."""
        pl = self._plan
        return [float(-pl['hyps'][slot][r][0]) * (1 - self.param_groups[gi]['betas'][0] ** pl['count'][r])
                for r, gi in enumerate(pl['groups'])]
