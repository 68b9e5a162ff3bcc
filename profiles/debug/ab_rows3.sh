ENS="c5" CHECK= bash profiles/quick_ens.sh 2>&1 | grep -E "^==|rep 2|grid_rows"
KERNEL='grid_rows_kernel<2>' bash profiles/pmc_rows.sh 2>&1 | grep -E "^==|per launch|LDS|VALU|WAVE_CYCLES"
KERNEL='grid_rows_kernel<1>' bash profiles/pmc_rows.sh 2>&1 | grep -E "^==|per launch|LDS|VALU|WAVE_CYCLES"
