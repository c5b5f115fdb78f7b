"""The regression refinement block (``BasicUpdateBlock``, /root/reference/unimatch/reg_refine.py:79-119) in channels-last
layout on the library's matrix-core convolutions -- SURVEY.md 8(f) rank 3.

Same parameters and arithmetic as :class:`unimatch_amd.refine.BasicUpdateBlock` (which stays the CPU / reference path);
what changes is the data flow:

* the local cost volume (K4, ``um_local_corr_with_flow_planes``) is written directly as the operand planes of the motion
  encoder's 1x1 convolution -- the fp32 ``[B, 81, h, w]`` volume never exists;
* every convolution (``um_conv2d_ex`` / ``um_conv7_fwd``) writes the NEXT convolution's operand planes from its epilogue,
  with ReLU / sigmoid / tanh fused; the four ``torch.cat`` of the block are column offsets into shared buffers:
      CF  [256] = convc2 out (192) | convf2 out (64)
      G   [512] = inp (128) | h (128) | motion (128 - fd) | flow (fd) | r * h (128)
  so ``hx = cat(h, x)`` is columns 0..384 of G and ``cat(r * h, x)`` is columns 0..128 + 256..512, read with the gate weights'
  input channels permuted to G's order;
* round 4 -- the iteration-invariant share of the GRU's convolutions is computed ONCE per scale: the loop restarts the hidden
  state from the same ``net0`` and feeds the same context ``inp`` in every iteration (unimatch.py:315-331), and a convolution is
  linear in its input channels, so  conv(W, [net0 | inp | motion | flow]) = conv(W[:, net0 | inp], .) + conv(W[:, motion | flow], .):
  ``begin()`` evaluates the first term of the four gate convolutions (pass 1: z|r over inp and net0, q over inp; pass 2: z|r and q
  over inp; bias included, no activation) into fp32 tables, and the per-iteration gate convolutions read only the columns that
  change -- 128 instead of 384 input channels for z|r of pass 1, 256 instead of 384 for the other three -- and take the table as
  the epilogue's addend (``um_conv2d_gru_add_fwd``).  For n iterations that removes (n - 1) / n x 25 % of the block's convolution
  FLOPs (config 4, n = 6: 21 %); with one iteration nothing is hoisted;
* the z and r gates are one convolution (256 outputs, sigmoid epilogue) whose epilogue also forms ``r * h``; the q
  convolution's epilogue (tanh) performs the state update ``h <- (1 - z) h + z q`` (``um_conv2d_gru_fwd``);
* the mask head only runs in the iteration whose mask is used (the reference computes and discards the others).
"""
import torch


class NhwcUpdateBlock:
    def __init__(self, ops, block, proj):
        self.ops, self.block, self.proj = ops, block, proj
        self.fd = block.flow_head.conv2.out_channels                 # flow channels: 2 (flow) or 1 (disparity / depth)

    # ---------------------------------------------------------------- weights (prepared once per parameter version)
    def _w(self, tag, params, build):
        key, hit = self.ops._cache_get(('refine', tag), tuple(params))
        if hit is None:
            with torch.no_grad():
                w, b = build()
                planes = self.ops.conv_weight_planes_from(w)
                hit = self.ops._cache_put(key, tuple(params), (planes, None if b is None else b.float().contiguous()))
        return hit

    @staticmethod
    def _pad(t, dim, to):
        if t.shape[dim] == to:
            return t
        shape = list(t.shape)
        shape[dim] = to - t.shape[dim]
        return torch.cat([t, t.new_zeros(shape)], dim)

    def _weights(self):
        enc, gru, fh, fd = self.block.encoder, self.block.gru, self.block.flow_head, self.fd
        w = {}
        w['proj'] = self._w('proj', [self.proj.weight, self.proj.bias], lambda: (self.proj.weight, self.proj.bias))
        w['c1'] = self._w('c1', [enc.convc1.weight, enc.convc1.bias],
                          lambda: (self._pad(enc.convc1.weight, 1, 96), enc.convc1.bias))
        w['c2'] = self._w('c2', [enc.convc2.weight, enc.convc2.bias], lambda: (enc.convc2.weight, enc.convc2.bias))
        w['f2'] = self._w('f2', [enc.convf2.weight, enc.convf2.bias], lambda: (enc.convf2.weight, enc.convf2.bias))
        w['mo'] = self._w('mo', [enc.conv.weight, enc.conv.bias],
                          lambda: (self._pad(enc.conv.weight, 0, 128), self._pad(enc.conv.bias, 0, 128)))
        # gate convolutions: input channels of the reference are hx = (h | inp | motion+flow) for z / r and (r*h | inp | motion+flow) for
        # q.  Hoisted form (G = inp | h | motion+flow | r*h): '*i' = the iteration-invariant columns (bias included there), '*v' = the
        # columns that change.  'zr*' / 'q*': the whole convolution (one iteration: nothing to hoist).
        hsl, isl, msl = slice(0, 128), slice(128, 256), slice(256, 384)
        for tag in ('1', '2'):
            z, r, q = (getattr(gru, f'conv{g}{tag}') for g in 'zrq')
            zrw = lambda z=z, r=r: torch.cat([z.weight, r.weight], 0)
            zrb = lambda z=z, r=r: torch.cat([z.bias, r.bias], 0)
            ps = [z.weight, z.bias, r.weight, r.bias]
            # not hoisted (one iteration; G = h | inp | motion+flow | r*h): z|r over columns 0..384 as they are, q over 128..512
            w['zr' + tag] = self._w('zr' + tag, ps, lambda zrw=zrw, zrb=zrb: (zrw(), zrb()))
            w['q' + tag] = self._w('q' + tag, [q.weight, q.bias],
                                   lambda q=q: (torch.cat([q.weight[:, 128:], q.weight[:, :128]], 1), q.bias))
            if tag == '1':      # pass 1: h = net0 is invariant too
                w['zr1i'] = self._w('zr1i', ps, lambda zrw=zrw, zrb=zrb: (torch.cat([zrw()[:, isl], zrw()[:, hsl]], 1), zrb()))
                w['zr1v'] = self._w('zr1v', ps, lambda zrw=zrw: (zrw()[:, msl].contiguous(), None))
            else:
                w['zr2i'] = self._w('zr2i', ps, lambda zrw=zrw, zrb=zrb: (zrw()[:, isl].contiguous(), zrb()))
                w['zr2v'] = self._w('zr2v', ps, lambda zrw=zrw: (torch.cat([zrw()[:, hsl], zrw()[:, msl]], 1), None))
            w['q' + tag + 'i'] = self._w('q' + tag + 'i', [q.weight, q.bias], lambda q=q: (q.weight[:, isl].contiguous(), q.bias))
            w['q' + tag + 'v'] = self._w('q' + tag + 'v', [q.weight, q.bias],
                                         lambda q=q: (torch.cat([q.weight[:, msl], q.weight[:, hsl]], 1), None))
        w['fh1'] = self._w('fh1', [fh.conv1.weight, fh.conv1.bias], lambda: (fh.conv1.weight, fh.conv1.bias))
        w['fh2'] = self._w('fh2', [fh.conv2.weight, fh.conv2.bias],
                           lambda: (self._pad(fh.conv2.weight, 0, 4), self._pad(fh.conv2.bias, 0, 4)))
        if self.block.mask is not None:
            m1, m2 = self.block.mask[0], self.block.mask[2]
            w['m1'] = self._w('m1', [m1.weight, m1.bias], lambda: (m1.weight, m1.bias))
            w['m2'] = self._w('m2', [m2.weight, m2.bias], lambda: (m2.weight, m2.bias))
        return w

    # ---------------------------------------------------------------- per scale
    def begin(self, f0_tokens, b, h, w, iterations=1):
        """``f0_tokens [b, h*w, 128]``: the (transformer) features the block's hidden state / context are projected from;
        ``iterations``: how often :meth:`iterate` will run (more than once: the invariant share of the gates is hoisted)."""
        ops = self.ops
        self.b, self.h, self.w = b, h, w
        rows = self.rows = b * h * w
        self.wts = self._weights()
        dev = f0_tokens.device
        self.G = ops.cached_planes_buffer('G', rows, 512, dev)
        self.C1, self.CF, self.FH = (ops.cached_planes_buffer(t, rows, 256, dev) for t in ('C1', 'CF', 'FH'))
        self.F1 = ops.cached_planes_buffer('F1', rows, 128, dev)
        self.CORR = ops.cached_planes_buffer('CORR', rows, 96, dev)
        fp, _ = ops.nhwc_planes_from([f0_tokens.reshape(rows, 128)])
        proj = torch.empty((rows, 256), dtype=torch.float32, device=f0_tokens.device)
        ops.conv_ex((fp, 128, 0, 128), (b, h, w), self.wts['proj'], (1, 1), 1, (0, 0), 0, out=(proj, 256, 0))
        self.net0 = torch.tanh(proj[:, :128]).contiguous()            # unimatch.py:317-320
        inp = torch.relu(proj[:, 128:]).contiguous()
        self.hoist = iterations > 1 and getattr(ops, 'refine_hoist', True)
        # G's columns: hoisted  inp 0 | h 128 | motion+flow 256 | r*h 384   (the changing columns are contiguous for every gate);
        #              plain    h 0 | inp 128 | motion+flow 256 | r*h 384   (the reference's own order)
        self.c_inp, self.c_h = (0, 128) if self.hoist else (128, 0)
        ops.nhwc_gate(0, inp, self.G, 512, self.c_inp, rows, 128)
        self.H = torch.empty_like(self.net0)
        self.ZR = torch.empty((rows, 256), dtype=torch.float32, device=proj.device)
        self.D = torch.empty((rows, 4), dtype=torch.float32, device=proj.device)
        if self.hoist:
            g, W = (b, h, w), self.wts
            ops.nhwc_gate(0, self.net0, self.G, 512, 128, rows, 128)  # h = net0 at the start of every iteration
            new = lambda c: torch.empty((rows, c), dtype=torch.float32, device=proj.device)
            self.P = {'zr1': new(256), 'q1': new(128), 'zr2': new(256), 'q2': new(128)}
            ops.conv_ex((self.G, 512, 0, 256), g, W['zr1i'], (1, 5), 1, (0, 2), 0, out=(self.P['zr1'], 256, 0))
            ops.conv_ex((self.G, 512, 0, 128), g, W['q1i'], (1, 5), 1, (0, 2), 0, out=(self.P['q1'], 128, 0))
            ops.conv_ex((self.G, 512, 0, 128), g, W['zr2i'], (5, 1), 1, (2, 0), 0, out=(self.P['zr2'], 256, 0))
            ops.conv_ex((self.G, 512, 0, 128), g, W['q2i'], (5, 1), 1, (2, 0), 0, out=(self.P['q2'], 128, 0))

    def iterate(self, ori0, ori1, disp, flow, want_mask):
        """One refinement iteration.  ``disp [b,2,h,w]``: sampling offsets of the cost volume, ``flow [b,fd,h,w]``: the
        current estimate.  Returns ``(mask_nhwc | None, delta [b,fd,h,w])``."""
        ops, b, h, w, rows, fd, W = self.ops, self.b, self.h, self.w, self.rows, self.fd, self.wts
        g = (b, h, w)
        ops.local_corr_with_flow_planes(ori0, ori1, disp, h, w, 4, self.CORR, 96)
        # motion encoder (reg_refine.py:6-36)
        ops.conv_ex((self.CORR, 96, 0, 96), g, W['c1'], (1, 1), 1, (0, 0), 1, outp=(self.C1, 256, 0))
        ops.conv_ex((self.C1, 256, 0, 256), g, W['c2'], (3, 3), 1, (1, 1), 1, outp=(self.CF, 256, 0))
        enc = self.block.encoder
        ops.conv7(flow, enc.convf1.weight, enc.convf1.bias, 1, 1, outp=(self.F1, 128, 0))
        ops.conv_ex((self.F1, 128, 0, 128), g, W['f2'], (3, 3), 1, (1, 1), 1, outp=(self.CF, 256, 192))
        ops.conv_ex((self.CF, 256, 0, 256), g, W['mo'], (3, 3), 1, (1, 1), 1, outp=(self.G, 512, 256))
        ops.nhwc_gate(0, flow.permute(0, 2, 3, 1).reshape(rows, fd).contiguous(), self.G, 512, 384 - fd, rows, fd)
        # SepConvGRU (reg_refine.py:55-76); the hidden state restarts from net0 every iteration (unimatch.py:322-331)
        if not self.hoist:
            self.H.copy_(self.net0)
            ops.nhwc_gate(0, self.H, self.G, 512, self.c_h, rows, 128)
        for tag, ks, pad in (('1', (1, 5), (0, 2)), ('2', (5, 1), (2, 0))):
            # gate arithmetic in the convolutions' epilogues: (z | r) -> z (fp32) and r * h (planes); q -> h updated in place
            if self.hoist:      # only the columns that change; the invariant share comes in as the epilogue's addend
                # pass 1 reads net0 itself (its planes are not read at all: h's share of z | r sits in the addend) and the first q
                # convolution writes the working state H and its planes: no copy of net0, no plane scatter per iteration (round 5)
                first = tag == '1'
                zsrc = (self.G, 512, 256, 128) if first else (self.G, 512, 128, 256)
                ops.conv_gru(1, zsrc, g, W['zr' + tag + 'v'], ks, pad, self.net0 if first else self.H, (self.G, 512, 384), z_out=self.ZR,
                             addend=self.P['zr' + tag])
                ops.conv_gru(2, (self.G, 512, 256, 256), g, W['q' + tag + 'v'], ks, pad, self.net0 if first else self.H, (self.G, 512, 128),
                             z=self.ZR, addend=self.P['q' + tag], hidden_out=self.H if first else None)
            else:
                ops.conv_gru(1, (self.G, 512, 0, 384), g, W['zr' + tag], ks, pad, self.H, (self.G, 512, 384), z_out=self.ZR)
                ops.conv_gru(2, (self.G, 512, 128, 384), g, W['q' + tag], ks, pad, self.H, (self.G, 512, 0), z=self.ZR)
        # flow head (reg_refine.py:39-52)
        ops.conv_ex((self.G, 512, self.c_h, 128), g, W['fh1'], (3, 3), 1, (1, 1), 1, outp=(self.FH, 256, 0))
        ops.conv_ex((self.FH, 256, 0, 256), g, W['fh2'], (3, 3), 1, (1, 1), 0, out=(self.D, 4, 0))
        delta = self.D[:, :fd].reshape(b, h, w, fd).permute(0, 3, 1, 2).contiguous()
        mask = None
        if want_mask and self.block.mask is not None:
            ops.conv_ex((self.G, 512, self.c_h, 128), g, W['m1'], (3, 3), 1, (1, 1), 1, outp=(self.FH, 256, 0))
            cm = self.block.mask[2].out_channels
            mask = torch.empty((rows, cm), dtype=torch.float32, device=self.D.device)
            ops.conv_ex((self.FH, 256, 0, 256), g, W['m2'], (1, 1), 1, (0, 0), 0, out=(mask, cm, 0))
        return mask, delta
