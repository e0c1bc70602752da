"""Readable names for this library's kernels as rocprofv3 / the ELF notes print them.  Kernels whose arguments are plain
pointers come out mangled (and binutils' c++filt does not know _Float16):
    _ZN2q313linear_kernelILi2ELi2ELi4ELi8ELi1ELi2ELb1EEEvPKDF16_...   -> linear_kernel<2, 2, 4, 8, 1, 2, true>
    void q3::conv_kernel<4, 7, 8, false>(q3::ConvArgs)                  -> conv_kernel<4, 7, 8, false>"""
import re


def pretty(name: str) -> str:
    name = name[:-3] if name.endswith(".kd") else name
    m = re.match(r"_ZN2q3(\d+)", name)
    if m:
        n = int(m.group(1))
        base = name[m.end():m.end() + n]
        rest = name[m.end() + n:]
        t = re.match(r"I((?:L[ib]\d+E)+)E", rest)
        if t:
            args = [("true" if v == "1" else "false") if k == "b" else v for k, v in re.findall(r"L([ib])(\d+)E", t.group(1))]
            return f"{base}<{', '.join(args)}>"
        return base
    name = re.sub(r"\(.*$", "", name)
    return name.replace("void ", "").replace("q3::", "").strip()
