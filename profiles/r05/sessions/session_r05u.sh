#!/bin/bash
# round 5, session u: SQ instruction counters of the finishers (k = 27 quarter: shipped kernel, the collapsing one, the skew leg) and kernel statistics of the k = 55 / k = 127 quarter legs
bash tools/gpu_session.sh r05u pmck:27:SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_WAVES:A=1 pmck:27:SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_WAVES:KMC_HIP_RANK_COLLAPSE=1 pmck:27:SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_WAVES:KMC_SYNTH_REPEATS=10000:2000:10 profk:55:A=1 profk:127:A=1
