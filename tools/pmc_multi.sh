#!/usr/bin/env bash
# Usage: tools/pmc_multi.sh <kernel-name-regex> <bench args...> : several PMC passes (memory pipeline) for one workload
pat="$1"; shift
i=0
for ctrs in "TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
            "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
            "TCC_BUSY_avr TCC_REQ_sum TCC_TAG_STALL_sum TCC_EA0_WRREQ_STALL_sum" \
            "TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_LATENCY_FIFO_FULL_sum" \
            "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_MULTI_MISS_sum" \
            "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i+1))
  tools/gpu_pmc.sh m$i "$ctrs" "$@" 2>&1 | grep -E "$pat"
done
