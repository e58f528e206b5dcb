#!/usr/bin/env python3
"""Extract the network weights and normalisation scalars of the trained encoder model that
ships with the reference (references/ObserverUnbalancedDisk_dt01_nab_4_SNR_30_e250) into a
plain .npz.

The file is a torch zip archive whose data.pkl needs deepSI classes to unpickle.  Nothing is
unpickled here: data.pkl is walked with pickletools.genops (opcode listing only) by a tiny
inert stack machine that records, for every persistent-id storage reference, its key, element
count and the (offset, shape, stride) triple that follows it, plus the attribute names on the
path to it.  The raw little-endian fp32 blobs archive/data/<key> are then read straight from
the zip.

Parameter order inside each simple_res_net follows named_parameters():
net_lin.weight, net_lin.bias, net_non_lin.net.{0,2,4}.{weight,bias}  -- the order
mpcUtil.py:11-29 of the reference relies on (params_list[0,1] = linear skip, [2+2i,3+2i] =
hidden layer i, last pair = output layer).

Run in the build container only (needs /root/reference):
    python tools/extract_weights.py [src] [dst.npz]
"""
import pickletools
import sys
import zipfile

import numpy as np

SRC = "/root/reference/references/ObserverUnbalancedDisk_dt01_nab_4_SNR_30_e250"
DST = "nmpc_amd/data/disc_snr30.npz"


class _Opaque:
    """Placeholder for any object the pickle would construct; holds attribute dict only."""

    def __init__(self, tag):
        self.tag = tag
        self.state = None
        self.items = []

    def __repr__(self):
        return "<%s>" % (self.tag,)


_MARK = object()


def scan(data):
    """Inert walk of a pickle opcode stream.  Returns the root placeholder object."""
    stack, memo = [], {}

    def pop_mark():
        out = []
        while True:
            v = stack.pop()
            if v is _MARK:
                break
            out.append(v)
        out.reverse()
        return out

    for op, arg, _pos in pickletools.genops(data):
        n = op.name
        if n in ("PROTO", "FRAME"):
            pass
        elif n == "STOP":
            break
        elif n == "MARK":
            stack.append(_MARK)
        elif n in ("GLOBAL", "STACK_GLOBAL"):
            if n == "STACK_GLOBAL":
                name = stack.pop(); mod = stack.pop(); arg = "%s %s" % (mod, name)
            stack.append(_Opaque("global:" + arg))
        elif n in ("BINPUT", "LONG_BINPUT", "PUT", "MEMOIZE"):
            memo[arg if n != "MEMOIZE" else len(memo)] = stack[-1]
        elif n in ("BINGET", "LONG_BINGET", "GET"):
            stack.append(memo[arg])
        elif n in ("BINUNICODE", "SHORT_BINUNICODE", "UNICODE", "BINUNICODE8",
                   "BININT", "BININT1", "BININT2", "LONG1", "LONG4", "INT", "LONG",
                   "BINFLOAT", "FLOAT", "BINBYTES", "SHORT_BINBYTES", "BINSTRING",
                   "SHORT_BINSTRING", "STRING"):
            stack.append(arg)
        elif n == "NONE":
            stack.append(None)
        elif n == "NEWTRUE":
            stack.append(True)
        elif n == "NEWFALSE":
            stack.append(False)
        elif n == "EMPTY_TUPLE":
            stack.append(())
        elif n == "EMPTY_LIST":
            stack.append([])
        elif n == "EMPTY_DICT":
            stack.append({})
        elif n == "EMPTY_SET":
            stack.append(set())
        elif n == "TUPLE":
            stack.append(tuple(pop_mark()))
        elif n == "TUPLE1":
            stack[-1:] = [(stack[-1],)]
        elif n == "TUPLE2":
            stack[-2:] = [(stack[-2], stack[-1])]
        elif n == "TUPLE3":
            stack[-3:] = [(stack[-3], stack[-2], stack[-1])]
        elif n == "LIST":
            stack.append(list(pop_mark()))
        elif n == "DICT":
            kv = pop_mark(); stack.append(dict(zip(kv[0::2], kv[1::2])))
        elif n == "APPEND":
            v = stack.pop(); stack[-1].append(v)
        elif n == "APPENDS":
            vs = pop_mark(); stack[-1].extend(vs)
        elif n == "SETITEM":
            v = stack.pop(); k = stack.pop(); _setitem(stack[-1], k, v)
        elif n == "SETITEMS":
            kv = pop_mark()
            for k, v in zip(kv[0::2], kv[1::2]):
                _setitem(stack[-1], k, v)
        elif n == "ADDITEMS":
            vs = pop_mark()
        elif n in ("NEWOBJ", "REDUCE"):
            args = stack.pop(); f = stack.pop()
            o = _Opaque("%s(%s)" % (n.lower(), getattr(f, "tag", f)))
            o.args = args
            stack.append(o)
        elif n == "NEWOBJ_EX":
            kw = stack.pop(); args = stack.pop(); f = stack.pop()
            o = _Opaque("newobj(%s)" % getattr(f, "tag", f)); o.args = args
            stack.append(o)
        elif n == "BUILD":
            st = stack.pop(); stack[-1].state = st
        elif n == "BINPERSID":
            pid = stack.pop()
            o = _Opaque("storage"); o.pid = pid
            stack.append(o)
        elif n == "POP":
            stack.pop()
        elif n == "POP_MARK":
            pop_mark()
        elif n == "DUP":
            stack.append(stack[-1])
        else:
            raise ValueError("unhandled pickle opcode %s" % n)
    return stack[-1]


def _setitem(container, k, v):
    if isinstance(container, dict):
        try:
            container[k] = v
        except TypeError:
            container[repr(k)] = v
    else:  # OrderedDict placeholder etc.
        container.items.append((k, v))


def _as_dict(o):
    """Attribute/items mapping of a placeholder (module __dict__ or OrderedDict content)."""
    if isinstance(o, dict):
        return o
    d = {}
    if isinstance(getattr(o, "state", None), dict):
        d.update(o.state)
    for k, v in getattr(o, "items", []):
        d[k] = v
    return d


def _tensor(zf, prefix, o):
    """o = reduce(_rebuild_parameter)(reduce(_rebuild_tensor_v2)(storage, offset, shape, stride, ...), ...)"""
    inner = o.args[0] if "rebuild_parameter" in o.tag else o
    storage, offset, shape, stride = inner.args[0], inner.args[1], inner.args[2], inner.args[3]
    _, stype, key, _dev, numel = storage.pid
    assert "FloatStorage" in stype.tag, stype.tag
    raw = np.frombuffer(zf.read("%s/data/%s" % (prefix, key)), dtype="<f4")
    assert raw.size == numel
    n = int(np.prod(shape)) if len(shape) else 1
    arr = raw[offset:offset + n].reshape(shape)
    exp_stride = tuple(int(np.prod(shape[i + 1:])) for i in range(len(shape)))
    assert tuple(stride) == exp_stride, (stride, exp_stride)
    return np.array(arr, dtype=np.float32), int(key)


def _net_params(zf, prefix, net_holder):
    """net_holder: module with _modules['net'] = simple_res_net."""
    res = _as_dict(_as_dict(_as_dict(net_holder)["_modules"])["net"])
    mods = _as_dict(res["_modules"])
    lin = _as_dict(mods["net_lin"])
    out = []
    p = _as_dict(lin["_parameters"])
    out += [_tensor(zf, prefix, p["weight"]), _tensor(zf, prefix, p["bias"])]
    seq = _as_dict(_as_dict(_as_dict(mods["net_non_lin"])["_modules"])["net"])
    layers = _as_dict(seq["_modules"])
    for name in sorted(layers, key=int):
        p = _as_dict(_as_dict(layers[name]).get("_parameters", {}))
        if "weight" in p:
            out += [_tensor(zf, prefix, p["weight"]), _tensor(zf, prefix, p["bias"])]
    return out


def _f8(o):
    # numpy scalar: reduce(scalar)(dtype, encode(latin1-str, 'latin1'))
    enc = o.args[1]
    s = enc.args[0]
    return float(np.frombuffer(s.encode("latin1"), dtype="<f8")[0])


def main(src=SRC, dst=DST):
    zf = zipfile.ZipFile(src)
    pkl = [n for n in zf.namelist() if n.endswith("data.pkl")][0]
    prefix = pkl.rsplit("/", 1)[0]
    root = scan(zf.read(pkl))
    top = _as_dict(root)
    norm = _as_dict(top["norm"])
    out = {k: np.float64(_f8(norm[k])) for k in ("u0", "ustd", "y0", "ystd")}
    for k in ("nx", "na", "nb", "k0", "na_right", "nb_right"):
        out[k] = np.int64(top[k])
    names = ["lin_w", "lin_b", "w0", "b0", "w2", "b2", "w4", "b4"]
    keys = {}
    for net, attr in (("enc", "encoder"), ("fn", "fn"), ("hn", "hn")):
        params = _net_params(zf, prefix, top[attr])
        assert len(params) == 8, (net, len(params))
        for nm, (arr, key) in zip(names, params):
            out["%s_%s" % (net, nm)] = arr
            keys["%s_%s" % (net, nm)] = key
    np.savez(dst, **out)
    for k in sorted(out):
        v = out[k]
        print("%-12s %-10s %s" % (k, getattr(v, "shape", ()), ("storage %d" % keys[k]) if k in keys else v))


if __name__ == "__main__":
    main(*sys.argv[1:])
