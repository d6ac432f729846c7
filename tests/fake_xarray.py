"""A minimal stand-in for the parts of xarray that gcm_filters' Filter touches (xarray is not installed in
this image).  TEST-ONLY.  It models the semantics the adapter relies on -- listed in README.md ("xarray semantics assumed") so that the
first user with real xarray knows where to look:
  * DataArray(data, dims) / Dataset(data_vars, coords) / ds.copy(deep) / ds.variables / ds[key] = da / ds.name / da.dims / da.dtype /
    da.mean(dim=[...]) / testing.assert_allclose -- what upstream tests/test_filter.py:172-252 uses;
  * apply_ufunc(func, *args, input_core_dims, output_core_dims, output_dtypes, dask): exactly upstream's keyword set
    (reference filter.py:478-486, 518-527); restated from xarray's documented behaviour (core/computation.py: apply_variable_ufunc,
    broadcast_compat_data, unified_dim_sizes):
      - every operand must HAVE all of its core dims (ValueError otherwise) and may not repeat a dim;
      - broadcast dims = the non-core dims of all operands in order of first appearance; sizes must agree by NAME (ValueError);
      - an operand reaches `func` as its data transposed to (its own broadcast dims in that order..., core dims...), with a length-1
        axis inserted for a broadcast dim it lacks ONLY to the right of its first own dim (numpy broadcasting handles leading ones):
        a (y, x) grid variable next to a (time, y, x) field arrives 2-D, a (time, y, x) variable next to a (time, z, y, x) field
        arrives as (time, 1, y, x);
      - every output must have len(broadcast dims) + len(its core dims) axes (ValueError otherwise) and gets those dims; sizes of
        known dims are checked;
      - dask="parallelized" needs `output_dtypes` (one per output) and calls `func` block by block from worker threads.
"""
import copy

import numpy as np


class DataArray:
    def __init__(self, data, dims=None, coords=None, name=None):
        if isinstance(data, DataArray):
            dims, data = data.dims, data.data
        self.data = np.asarray(data)
        self.dims = tuple(dims) if dims is not None else tuple(f"dim_{k}" for k in range(self.data.ndim))
        assert len(self.dims) == self.data.ndim
        self.name = name

    values = property(lambda self: self.data)
    dtype = property(lambda self: self.data.dtype)
    shape = property(lambda self: self.data.shape)
    ndim = property(lambda self: self.data.ndim)

    def transpose(self, *dims):
        order = [self.dims.index(d) for d in dims]
        return DataArray(self.data.transpose(order), dims)

    def _binary(self, other, op):
        if isinstance(other, DataArray):
            assert other.dims == self.dims[-other.data.ndim:] or other.dims == self.dims
            other = other.data
        return DataArray(op(self.data, other), self.dims)

    def __mul__(self, o):
        return self._binary(o, np.multiply)

    def __pow__(self, o):
        return DataArray(self.data ** o, self.dims)

    def sum(self):
        return self.data.sum()

    def mean(self, dim):
        dim = [dim] if isinstance(dim, str) else list(dim)
        ax = tuple(self.dims.index(d) for d in dim)
        return DataArray(self.data.mean(axis=ax), [d for d in self.dims if d not in dim])

    def __array__(self, dtype=None, copy=None):
        return self.data if dtype is None else self.data.astype(dtype)


Variable = DataArray


class Dataset:
    def __init__(self, data_vars=None, coords=None):
        self._vars = {}
        for k, v in (data_vars or {}).items():
            if isinstance(v, tuple):
                v = DataArray(v[1], v[0])
            self._vars[k] = v if isinstance(v, DataArray) else DataArray(v)
        self.coords = dict(coords or {})

    @property
    def variables(self):
        return dict(self._vars)

    def copy(self, deep=False):
        return copy.deepcopy(self) if deep else Dataset(self._vars, self.coords)

    def __getitem__(self, k):
        return self._vars[k]

    def __setitem__(self, k, v):
        self._vars[k] = v

    def __getattr__(self, k):
        try:
            return self.__dict__["_vars"][k]
        except KeyError:
            raise AttributeError(k)


def chunked(da, dim, nchunks):
    """Mark `da` as lazily chunked along the non-core dim `dim` (a stand-in for ``da.chunk({dim: ...})``): apply_ufunc with
    dask="parallelized" then calls the function once per block, CONCURRENTLY from a pool of worker threads -- the way
    dask's threaded scheduler drives ``filter_func`` (reference filter.py:485, docs/basic_filtering.rst:175-203)."""
    out = DataArray(da.data, da.dims, name=da.name)
    out._chunks = (dim, int(nchunks))
    return out


def _apply_blockwise(func, args, input_core_dims, output_core_dims, dim, nchunks):
    from concurrent.futures import ThreadPoolExecutor

    n = [a for a in args if dim in a.dims][0].data.shape[[a for a in args if dim in a.dims][0].dims.index(dim)]
    edges = np.linspace(0, n, min(nchunks, n) + 1).astype(int)

    def block(lo, hi):
        sub = []
        for a in args:
            if dim in a.dims:
                idx = [slice(None)] * a.data.ndim
                idx[a.dims.index(dim)] = slice(lo, hi)
                sub.append(DataArray(a.data[tuple(idx)], a.dims))
            else:
                sub.append(a)
        return apply_ufunc(func, *sub, input_core_dims=input_core_dims, output_core_dims=output_core_dims)

    with ThreadPoolExecutor(max_workers=4) as pool:
        parts = list(pool.map(lambda e: block(*e), zip(edges[:-1], edges[1:])))
    multi = len(output_core_dims) > 1
    cols = list(zip(*parts)) if multi else [parts]
    outs = tuple(DataArray(np.concatenate([p.data for p in col], axis=col[0].dims.index(dim)), col[0].dims) for col in cols)
    return outs if multi else outs[0]


def apply_ufunc(func, *args, input_core_dims, output_core_dims, output_dtypes=None, dask=None):
    if len(input_core_dims) != len(args):
        raise ValueError(f"input_core_dims must have one entry per argument: {len(input_core_dims)} for {len(args)} arguments")
    if dask not in (None, "forbidden", "allowed", "parallelized"):
        raise ValueError(f"unknown setting for dask array handling in apply_ufunc: {dask}")
    if output_dtypes is not None and len(output_dtypes) != len(output_core_dims):
        raise ValueError("output_dtypes must have one entry per output")
    lazy = [a._chunks for a in args if getattr(a, "_chunks", None)]
    if lazy and dask == "parallelized":
        if output_dtypes is None:
            raise ValueError("output dtypes (output_dtypes) must be supplied to apply_ufunc when using dask='parallelized'")
        return _apply_blockwise(func, args, input_core_dims, output_core_dims, *lazy[0])
    if lazy and dask in (None, "forbidden"):
        raise ValueError("apply_ufunc encountered a chunked array on an argument, but handling for chunked arrays has not been enabled")
    # ---- dimension bookkeeping by NAME ----
    all_core = {d for core in input_core_dims for d in core} | {d for core in output_core_dims for d in core}
    sizes, broadcast_dims = {}, []
    for a, core in zip(args, input_core_dims):
        if len(set(a.dims)) != len(a.dims):
            raise ValueError(f"broadcasting cannot handle duplicate dimensions on a variable: {list(a.dims)}")
        missing = [d for d in core if d not in a.dims]
        if missing:
            raise ValueError(f"operand to apply_ufunc has required core dimensions {list(core)}, but some of these dimensions are "
                             f"absent on an input variable: {missing}")
        for d, n in zip(a.dims, a.data.shape):
            if d in sizes and sizes[d] != n:
                raise ValueError(f"operands cannot be broadcast together with mismatched lengths for dimension {d!r}: {(sizes[d], n)}")
            sizes.setdefault(d, n)
            if d not in all_core and d not in broadcast_dims:
                broadcast_dims.append(d)
    raw = []
    for a, core in zip(args, input_core_dims):
        core = list(core)
        unexpected = [d for d in a.dims if d not in broadcast_dims and d not in core]
        if unexpected:
            raise ValueError(f"operand to apply_ufunc encountered unexpected dimensions {unexpected} on an input variable: these are "
                             "core dimensions on other input or output variables")
        own = [d for d in broadcast_dims if d in a.dims]
        data = a.transpose(*own, *core).data
        key, seen = [], False
        for d in broadcast_dims + core:
            if d in a.dims:
                key.append(slice(None))
                seen = True
            elif seen:              # (leading axes are left to numpy's broadcasting)
                key.append(np.newaxis)
        raw.append(data[tuple(key)] if len(key) != data.ndim else data)
    res = func(*raw)
    multi = len(output_core_dims) > 1
    if multi and not (isinstance(res, tuple) and len(res) == len(output_core_dims)):
        raise ValueError(f"applied function does not have the number of outputs specified in the ufunc signature: {len(output_core_dims)}")
    res = res if multi else (res,)
    outs = []
    for r, core in zip(res, output_core_dims):
        dims = tuple(broadcast_dims) + tuple(core)
        r = np.asarray(r)
        if r.ndim != len(dims):
            raise ValueError(f"applied function returned data with an unexpected number of dimensions. Received {r.ndim} dimension(s) "
                             f"but expected {len(dims)} dimensions with names {dims!r}")
        for d, n in zip(dims, r.shape):
            if d in sizes and sizes[d] != n:
                raise ValueError(f"size of dimension {d!r} on inputs was unexpectedly changed by applied function from {sizes[d]} to {n}")
        outs.append(DataArray(r, dims))
    return tuple(outs) if multi else outs[0]


class testing:   # noqa: N801  (xarray.testing)
    @staticmethod
    def assert_allclose(a, b, rtol=1e-5, atol=1e-8):
        assert tuple(a.dims) == tuple(b.dims), (a.dims, b.dims)
        np.testing.assert_allclose(np.asarray(a.data), np.asarray(b.data), rtol=rtol, atol=atol)
