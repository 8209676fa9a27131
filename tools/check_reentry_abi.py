#!/usr/bin/env python
"""Build-time check for k_fit2x (scarlet_amd/csrc/fused2.h).

k_fit2x ends an iteration by jumping back to its own first instruction with the launch registers re-created by
hand.  Which registers those are is fixed by the HSA ABI as a function of the kernel descriptor the compiler
emitted; this script reads that descriptor out of the built library and fails the build unless it says exactly
what the jump assumes:

    user SGPRs          = 2: the kernel-argument pointer in s[0:1] and nothing else
    system SGPRs        = workgroup id x, y, z in s2, s3, s4; no workgroup-info register
    private segment     = none (no scratch: no flat-scratch / wave-offset registers)
    work-item id VGPRs  = v0 only

    python tools/check_reentry_abi.py scarlet_amd/csrc/libscarlet_hip.so
"""
import os
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
KERNEL = "k_fit2x"


def elf_symbol_bytes(path, name, size):
    """bytes of the object symbol `name` in the ELF64 file at `path`"""
    data = open(path, "rb").read()
    assert data[:4] == b"\x7fELF" and data[4] == 2, "not an ELF64 file"
    shoff, = struct.unpack_from("<Q", data, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", data, 0x3A)
    secs = [struct.unpack_from("<IIQQQQIIQQ", data, shoff + i * shentsize) for i in range(shnum)]
    for sec in secs:
        if sec[1] != 2:                      # SHT_SYMTAB
            continue
        stroff = secs[sec[6]][4]
        for off in range(sec[4], sec[4] + sec[5], 24):
            st_name, _info, _other, shndx, value, _sz = struct.unpack_from("<IBBHQQ", data, off)
            end = data.index(b"\0", stroff + st_name)
            if data[stroff + st_name:end].decode() == name:
                s = secs[shndx]
                fo = s[4] + (value - s[3])
                return data[fo:fo + size]
    raise SystemExit("check_reentry_abi: symbol %s not found" % name)


def main():
    so = sys.argv[1]
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, so])
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--unbundle",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co])
        kd = elf_symbol_bytes(co, KERNEL + ".kd", 64)
    private_size, = struct.unpack_from("<I", kd, 4)
    rsrc2, = struct.unpack_from("<I", kd, 52)
    props, preload = struct.unpack_from("<HH", kd, 56)
    got = dict(private_segment_fixed_size=private_size,
               enable_private_segment=rsrc2 & 1,
               user_sgpr_count=(rsrc2 >> 1) & 0x1F,
               workgroup_id_x=(rsrc2 >> 7) & 1, workgroup_id_y=(rsrc2 >> 8) & 1, workgroup_id_z=(rsrc2 >> 9) & 1,
               workgroup_info=(rsrc2 >> 10) & 1, workitem_id_vgprs=(rsrc2 >> 11) & 3,
               sgpr_private_segment_buffer=props & 1, sgpr_dispatch_ptr=(props >> 1) & 1, sgpr_queue_ptr=(props >> 2) & 1,
               sgpr_kernarg_segment_ptr=(props >> 3) & 1, sgpr_dispatch_id=(props >> 4) & 1,
               sgpr_flat_scratch_init=(props >> 5) & 1, sgpr_private_segment_size=(props >> 6) & 1,
               wavefront_size32=(props >> 10) & 1, uses_dynamic_stack=(props >> 11) & 1,
               kernarg_preload=preload)
    want = dict(private_segment_fixed_size=0, enable_private_segment=0, user_sgpr_count=2,
                workgroup_id_x=1, workgroup_id_y=1, workgroup_id_z=1, workgroup_info=0, workitem_id_vgprs=0,
                sgpr_private_segment_buffer=0, sgpr_dispatch_ptr=0, sgpr_queue_ptr=0, sgpr_kernarg_segment_ptr=1,
                sgpr_dispatch_id=0, sgpr_flat_scratch_init=0, sgpr_private_segment_size=0, wavefront_size32=0,
                uses_dynamic_stack=0, kernarg_preload=0)
    bad = {k: (got[k], want[k]) for k in want if got[k] != want[k]}
    if bad:
        print("check_reentry_abi: %s's kernel descriptor is not what its re-entry jump assumes (got, wanted): %r"
              % (KERNEL, bad), file=sys.stderr)
        sys.exit(1)
    print("check_reentry_abi: %s descriptor ok (kernarg ptr s[0:1], workgroup ids s2-s4, v0, no scratch)" % KERNEL)


if __name__ == "__main__":
    main()
