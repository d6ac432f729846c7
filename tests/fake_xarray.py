"""A minimal stand-in for the parts of xarray that gcm_filters' Filter touches (xarray is not installed in
this image).  TEST-ONLY.  It models the documented semantics the adapter relies on:
  * DataArray(data, dims) / Dataset(dict) / ds.copy(deep) / ds.variables / ds[key] = da / da.dims / da.dtype
  * apply_ufunc(func, *args, input_core_dims, output_core_dims): core dims are moved to the END of every
    input, the function is called on the raw arrays, outputs get dims (broadcast dims..., *core dims).
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
    lazy = [a._chunks for a in args if getattr(a, "_chunks", None)]
    if lazy and dask == "parallelized":
        return _apply_blockwise(func, args, input_core_dims, output_core_dims, *lazy[0])
    raw, lead_dims = [], ()
    for a, core in zip(args, input_core_dims):
        core = list(core)
        other = [d for d in a.dims if d not in core]
        raw.append(a.transpose(*other, *core).data)
        if len(other) > len(lead_dims):
            lead_dims = tuple(other)
    res = func(*raw)
    multi = len(output_core_dims) > 1
    res = res if multi else (res,)
    outs = tuple(DataArray(r, lead_dims + tuple(core)) for r, core in zip(res, output_core_dims))
    return outs if multi else outs[0]
