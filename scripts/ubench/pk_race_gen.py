"""Generates bisect variants of scripts/ubench/pk_race.hip: `python pk_race_gen.py NAME N i j k ...` writes
pk_race_NAME.hip with `s_nop N` inserted behind instructions i, j, k ... of the packed chain (indices in the listing below)
and builds libpk_race_NAME.so."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CHAIN = [
    "v_mov_b32_e32 v135, v134",
    "v_pk_add_f32 v[158:159], v[182:183], v[134:135] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]",
    "v_pk_add_f32 v[234:235], v[180:181], v[134:135] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]",
    "v_mul_f32_e32 v140, v159, v159",
    "v_pk_add_f32 v[240:241], v[142:143], v[134:135] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]",
    "v_pk_fma_f32 v[158:159], v[158:159], v[158:159], v[140:141] op_sel_hi:[1,1,0]",
    "v_mul_f32_e32 v140, v235, v235",
    "v_pk_add_f32 v[238:239], v[184:185], v[134:135] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]",
    "v_pk_mul_f32 v[240:241], v[240:241], v[240:241]",
    "v_pk_add_f32 v[158:159], v[140:141], v[158:159] op_sel_hi:[0,1]",
    "v_pk_add_f32 v[236:237], v[144:145], v[134:135] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]",
    "v_pk_fma_f32 v[238:239], v[238:239], v[238:239], v[240:241]",
    "v_pk_fma_f32 v[158:159], v[234:235], v[234:235], v[158:159]",
    "v_pk_add_f32 v[234:235], v[178:179], v[134:135] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]",
    "v_pk_fma_f32 v[236:237], v[236:237], v[236:237], v[238:239]",
    "s_nop 0",
    "v_pk_fma_f32 v[234:235], v[234:235], v[234:235], v[236:237]",
    "s_nop 0",
    "v_pk_add_f32 v[158:159], v[234:235], v[158:159]",
    "s_nop 0",
    "v_pk_add_f32 v[158:159], v[158:159], v[234:235] op_sel:[0,1] op_sel_hi:[1,0]",
]


def main():
    name, n = sys.argv[1], int(sys.argv[2])
    pads = set(int(a) for a in sys.argv[3:])
    body = []
    for i, ins in enumerate(CHAIN):
        body.append(f'            "{ins}\\n"')
        if i in pads:
            body.append(f'            "s_nop {n}\\n"')
    src = open(os.path.join(HERE, "pk_race.hip")).read()
    a = src.index('            "v_mov_b32_e32 v135, v134\\n"')
    b = src.index('            "s_nop 4\\n"\n            "v_mov_b32 %0, v158')
    out = src[:a] + "\n".join(body) + "\n" + src[b:]
    path = os.path.join(HERE, f"pk_race_{name}.hip")
    open(path, "w").write(out)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", "-w", path, "-o",
                           os.path.join(HERE, f"libpk_race_{name}.so")])
    print("built", name, "nop", n, "after", sorted(pads))


if __name__ == "__main__":
    main()
