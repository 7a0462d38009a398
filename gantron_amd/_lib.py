"""ctypes binding of libgantron_hip.so, generated from include/gantron_hip.h.

The C header is the single source of truth for the boundary: struct layouts and function
prototypes are parsed from it at import time, so the Python side cannot drift from the ABI.
Loading fails loudly when the shared object is missing: there is no CPU fallback.
"""
import ctypes
import os
import re

_PKG = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_PKG), 'include', 'gantron_hip.h')
LIB_PATH = os.path.join(_PKG, 'libgantron_hip.so')

_SCALARS = {
    'int32_t': ctypes.c_int32, 'int64_t': ctypes.c_int64, 'uint64_t': ctypes.c_uint64, 'uint32_t': ctypes.c_uint32,
    'float': ctypes.c_float, 'int': ctypes.c_int,
}


def _strip_comments(text):
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return re.sub(r'//[^\n]*', '', text)


def _parse_header(path):
    text = _strip_comments(open(path).read())
    structs = {}
    for m in re.finditer(r'typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;', text, flags=re.S):
        name, body = m.group(3), m.group(2)
        fields = []
        for decl in body.split(';'):
            decl = ' '.join(decl.split())
            if not decl:
                continue
            if '*' in decl:                       # pointer field(s): "const float* a" (one per declaration)
                fname = decl.replace('*', ' ').split()[-1]
                fields.append((fname, ctypes.c_void_p))
                continue
            parts = decl.replace(',', ' ').split()
            ctype, names = parts[0], parts[1:]
            typ = _SCALARS.get(ctype) or structs[ctype]
            fields.extend((n, typ) for n in names)
        structs[name] = type(name, (ctypes.Structure,), {'_fields_': fields})
    protos = {}
    body = re.sub(r'typedef\s+struct.*?\}\s*\w+\s*;', '', text, flags=re.S)
    for m in re.finditer(r'(const char\*|int64_t|int32_t|int)\s+(gt_\w+)\s*\(([^;]*?)\)\s*;', body, flags=re.S):
        ret, name, args = m.group(1), m.group(2), ' '.join(m.group(3).split())
        argtypes = []
        if args and args != 'void':
            for a in args.split(','):
                a = a.strip()
                if '*' in a:
                    base = a.replace('const', '').replace('*', ' ').split()[0]
                    argtypes.append(ctypes.POINTER(structs[base]) if base in structs else ctypes.c_void_p)
                else:
                    base = a.split()[0]
                    argtypes.append(_SCALARS.get(base) or structs[base])
        restype = {'const char*': ctypes.c_char_p, 'int64_t': ctypes.c_int64,
                   'int32_t': ctypes.c_int32, 'int': ctypes.c_int}[ret]
        protos[name] = (restype, argtypes)
    return structs, protos


STRUCTS, PROTOS = _parse_header(HEADER)
ABI_VERSION = int(re.search(r'#define\s+GT_ABI_VERSION\s+(\d+)', open(HEADER).read()).group(1))
globals().update(STRUCTS)


class GantronHipError(RuntimeError):
    pass


_lib = None


def lib():
    """The loaded shared object with typed prototypes (torch must already be imported so that the
    HIP runtime it ships is the one this library binds to)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GantronHipError(
                f'{LIB_PATH} is missing: build it with `python -m gantron_amd.csrc.build` '
                '(the HIP extension is mandatory; there is no CPU fallback)')
        import torch  # noqa: F401  (loads libamdhip64 first)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in PROTOS.items():
            fn = getattr(handle, name)        # AttributeError here = header/library drift
            fn.restype, fn.argtypes = restype, argtypes
        if handle.gt_version() != ABI_VERSION:
            raise GantronHipError(f'ABI mismatch: {LIB_PATH} was built as version {handle.gt_version()}, {HEADER} declares '
                                  f'GT_ABI_VERSION {ABI_VERSION}: rebuild it (python -m gantron_amd.csrc.build)')
        for which, sname in enumerate(('GtOperand', 'GtGemm', 'GtSlabs', 'GtAttnDims', 'GtDecoderFwd',
                                       'GtDecoderBwd', 'GtBiLstm', 'GtDecoderInfer', 'GtPlstm', 'GtPlstmBwd', 'GtRecTail',
                                       'GtCellBwdTail')):
            want, got = handle.gt_abi_sizeof(which), ctypes.sizeof(STRUCTS[sname])
            if want != got:
                raise GantronHipError(f'ABI mismatch: sizeof({sname}) is {want} in the library, {got} in the binding')
        _lib = handle
    return _lib


def check(rc, what=''):
    if rc != 0:
        msg = lib().gt_last_error_string().decode(errors='replace')
        raise GantronHipError(f'{what} failed with status {rc}: {msg}')
