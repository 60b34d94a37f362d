"""rocprofv3 kernel name -> the name the library's HIP-event timers (csrc/: ScopedTimer) report the same launch under.

The timers name a kernel as rocprofv3 does, without the `avid::` namespace and the argument list, and with fewer
template arguments where several instantiations are ONE kernel per layer shape to the host code that times them:
the table below says how many leading template arguments a timer name keeps.  tools/pmc_traffic.py keys
profiles/pmc_traffic.json with these names; tests/test_tools.py checks that every timer name of the committed bench
record is produced by a kernel of the committed rocprof summary (same round)."""
import re

# kernel -> number of leading template arguments in the timer name (absent: all of them)
KEEP = {
    "igemm_pk_kernel": 5,          # <WM,WN,TM,TN,MODE | STRIDED,EPI,BS>: STRIDED becomes the suffix "s2"
    "igemm_kernel": 5,             # <WM,WN,TM,TN,MODE | STRIDED>
    "tconv64_kernel": 1,           # <MODE | EPI, AFF (applies the BatchNorm in front of the layer)>
    "twgrad64_kernel": 0,          # <AFF>
    "stem_fwd3p_kernel": 2,        # <CIN,KT | TM (wave shape)>
    "stem_fwd_kernel": 2,          # <CIN,KT | WAVES>
    "stem_wgrad3_kernel": 2,       # <CIN,KT | PRE (dy fragments split by the loader)>
    "stem_wgrad_kernel": 2,        # <CIN,KT | wide K>
    "stem_wgrad_reduce_kernel": 2,
    "wgrad_group_kernel": 0,       # <SPLIT,PRE>
    "wgrad_tab_kernel": 2,         # <NB,KC | SPLIT>
    "wgrad_kernel": 2,
    "xmodal_fused_kernel": 0,      # <CMA,ROWS,NT>: CMA = true is timed as cma_fused_kernel
    "xmodal_finish_kernel": 0,
    "bank_scores_fwd_kernel": 0,   # <rows per wave>
    "bank_scores_bwd_kernel": 0,
    "splitk_reduce_cls_kernel": 0,
}
STRIDED_ARG = {"igemm_pk_kernel": 5, "igemm_kernel": 5}     # index of the bool that the timers write as the suffix "s2"


def timer_name(rocprof_name):
    """'void avid::igemm_pk_kernel<4, 1, 1, 2, 1, true, 9, true>(avid::ConvArgs)' -> 'igemm_pk_kernel<4,1,1,2,1>s2';
    None for kernels that are not the library's."""
    m = re.match(r"(?:void )?avid::(\w+)(?:<([^(]*)>)?", rocprof_name.strip())
    if not m:
        return None
    base, targs = m.group(1), m.group(2)
    if not targs:
        return base
    args = [a.strip() for a in targs.split(",")]
    suffix = ""
    if base in STRIDED_ARG and len(args) > STRIDED_ARG[base] and args[STRIDED_ARG[base]] == "true":
        suffix = "s2"
    if base in ("xmodal_fused_kernel", "xmodal_finish_kernel") and args[0] == "true":
        base = base.replace("xmodal", "cma")
    args = args[:KEEP.get(m.group(1), len(args))]
    return base + ("<" + ",".join(args) + ">" if args else "") + suffix
