"""NeRFRenderer.update_extra_state (renderer_wtmk.py:445-538) as device-side work without a host read, replayed as a hipGraph.

The trainer refreshes the density grid every 16 training steps (nerf/utils.py:852-858).  The reference's form synchronises with the host three times per
refresh (`nonzero` for the occupied cells, `.item()` for the mean density and for the mean sample count) and issues ~60 small tensor operations; between the
replays of a captured step that costs more than a step on a trained scene's sparse grid.  Here the refresh is a fixed sequence of launches on static buffers
(csrc/gridrefresh.hip: draw + counting sort -> points -> [encoder + sigma MLP] -> scatter per cascade, then EMA + mean + packbits + mean sample count), captured once per
form -- "full" for the first 16 refreshes, "partial" afterwards -- and replayed.  The values a host may ask for (mean_density, mean_count) stay on the device until
somebody reads the renderer's attribute.

The captured refresh holds libnerfsig kernel nodes and nothing else.  (A first version built the occupancy prefix with torch.cumsum and cleared its buffers with
fill_ / hipMemsetAsync inside the capture: the graph's SECOND replay, behind sixteen replays of the training step's graph, ended in "Memory access fault ... write access
to a read-only page" on this runtime -- reproducibly, and not with serialised launches.  The prefix sum, the fills and the counting sort are this library's own kernels now.)

Differences from the reference's form, all inside what it leaves undefined: the random numbers are a counter-based function of (seed, refresh count, cascade,
draw) instead of torch's generator; cells the partial refresh probes more than once keep the largest of their candidates (the reference: whichever write lands last).
The arithmetic of a probe point, of the EMA and of the threshold is the reference's, operation for operation."""
import torch

from . import _native as nv
from . import fieldops as fo


class DeviceGridRefresh:
    def __init__(self, model, seed=0, decay=0.95, capture=True):
        if not model.cuda_ray:
            raise ValueError("DeviceGridRefresh: the model has no occupancy grid (cuda_ray=False)")
        m = self.model = model
        dev = self.device = m.density_bitfield.device
        if dev.type != "cuda":
            raise RuntimeError("DeviceGridRefresh needs the grid on a GPU")
        self.seed, self.decay, self.capture = int(seed), float(decay), bool(capture)
        C, H = int(m.cascade), int(m.grid_size)
        self.C, self.H, self.cells = C, H, H ** 3
        self.n_draw = self.cells // 4                                # renderer_wtmk.py:488: N = H^3 / 4 uniform cells + N occupied ones per cascade
        cap = self.cells                                             # the full form probes every cell of a cascade
        i32, f32 = dict(dtype=torch.int32, device=dev), dict(dtype=torch.float32, device=dev)
        self.xyz = torch.empty(cap, 3, **f32)
        self.cell_index = torch.empty(cap, **i32)
        self.sigma = torch.empty(cap, **f32)
        self.keys = torch.empty(2 * self.n_draw, **i32)
        self.ids = torch.empty(2 * self.n_draw, **i32)
        self.draw_scratch = torch.empty(int(nv.fn("rg_refresh_draw_scratch_bytes")(self.n_draw, H)) // 4, **i32)
        self.fresh = torch.empty_like(m.density_grid)
        self.planes = torch.empty(int(nv.fn("hg_planes_bytes")(cap)), dtype=torch.uint8, device=dev)
        self.partials = torch.empty(int(nv.fn("rg_refresh_partials_bytes")(C * self.cells)) // 8, dtype=torch.float64, device=dev)
        self.iter_dev = torch.full((1,), int(m.iter_density), **i32)      # the refresh count: the random streams' counter
        self.mean_density_dev = torch.zeros(1, **f32)
        self.mean_count_dev = torch.zeros(1, **i32)
        self.graphs, self.seen = {}, set()
        self.stream = None

    def _probe(self, cas, n, keys, ids, packed):
        m, H = self.model, self.H
        extent, half_cell = m._cascade_extent(cas)
        s = nv.stream()
        nv.call("rg_refresh_points", nv.ptr(keys), nv.ptr(ids), n, H, float(extent), float(half_cell), self.seed, nv.ptr(self.iter_dev), cas, nv.ptr(self.xyz),
                nv.ptr(self.cell_index), s)
        base_ptrs = nv.ptr_array([t.detach() for t in m.encoder.tables()])
        layout = fo.encode_planes(self.xyz, n, m.bound, base_ptrs, None, self.planes)
        nv.call("field_fwd", nv.ptr(self.xyz), None, n, float(m.bound), base_ptrs, None, nv.ptr(packed), nv.ptr(self.sigma), None, None, None, nv.ptr(self.planes), layout, s)
        nv.call("rg_refresh_scatter", nv.ptr(self.sigma), nv.ptr(self.cell_index), n, float(m.density_scale), nv.ptr(self.fresh[cas]), s)

    def _body(self, form, window, count_ring, step_dev, packed):
        m = self.model
        nv.call("rg_refresh_begin", nv.ptr(self.fresh), self.C * self.cells, nv.stream())      # (libnerfsig launches only: the captured refresh holds kernel nodes, nothing else)
        for cas in range(self.C):
            if form == "full":
                self._probe(cas, self.cells, None, None, packed)
            else:
                nv.call("rg_refresh_draw", nv.ptr(self.keys), nv.ptr(self.ids), self.n_draw, self.H, nv.ptr(m.density_grid[cas]), nv.ptr(self.draw_scratch), self.seed,
                        nv.ptr(self.iter_dev), cas, nv.stream())      # (grouped by grid row: the order in which the encoder's gathers share lines)
                self._probe(cas, 2 * self.n_draw, self.keys, self.ids, packed)
        nv.call("rg_refresh_finish", nv.ptr(m.density_grid), nv.ptr(self.fresh), self.C * self.cells, self.decay, nv.ptr(self.partials), float(m.density_thresh),
                nv.ptr(m.density_bitfield), nv.ptr(self.mean_density_dev), nv.ptr(self.iter_dev), nv.ptr(count_ring) if window else None, int(window),
                nv.ptr(step_dev) if window else None, nv.ptr(self.mean_count_dev) if window else None, nv.stream())

    @torch.no_grad()
    def run(self, packed, count_ring=None, step_dev=None, window=0):
        """One refresh.  packed: the MLP weights' operand image (fieldops.pack_weights; a captured loop's static buffer); count_ring [16,2] int32 / step_dev: the loop's
        ring of (points, rays) per step and its device step count; window: how many of the ring's last rows the mean sample count averages (0: leave it alone).
        The first refresh of a form runs eagerly (it loads the kernels), the second is captured, later ones replay."""
        m = self.model
        if self.fresh.shape != m.density_grid.shape or m.density_grid.device != self.device:
            raise RuntimeError("DeviceGridRefresh: the model's grid changed shape or device since this object was built")
        form = "full" if m.iter_density < 16 else "partial"
        window = int(window) if (count_ring is not None and window) else 0
        key = (form, window, packed.data_ptr(), m.density_grid.data_ptr(), m.density_bitfield.data_ptr(), 0 if count_ring is None else count_ring.data_ptr())
        if key in self.graphs:
            self.graphs[key].replay()
        elif self.capture and key in self.seen:
            g = torch.cuda.CUDAGraph()
            if self.stream is None:
                self.stream = torch.cuda.Stream()
            torch.cuda.synchronize()
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                g.capture_begin(capture_error_mode="thread_local")
                try:
                    self._body(form, window, count_ring, step_dev, packed)
                finally:
                    g.capture_end()
            torch.cuda.current_stream().wait_stream(self.stream)
            self.graphs[key] = g
            g.replay()
        else:
            self.seen.add(key)
            self._body(form, window, count_ring, step_dev, packed)
        # the host-side books of update_extra_state (renderer_wtmk.py:525,537 and this repo's grid_key)
        m.iter_density += 1
        m._grid_epoch += 1
        m.local_step = 0
        m.__dict__["_mean_density_dev"] = self.mean_density_dev
        if window:
            m.__dict__["_mean_count_dev"] = self.mean_count_dev
