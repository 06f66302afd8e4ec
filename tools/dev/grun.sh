#!/bin/bash
# grun.sh [--timeout S] -- 'command'
# Wrapper around gpurun that KEEPS a record of every call of the round: the full client output goes to
# gpurun_out/r04_cNNN.call (scratch, merged like everything else there) and one line per call is appended to the tracked
# ledger profiles/r04_gpurun_ledger.md (status, rc, charged seconds, minutes left, the command).
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
n=$(ls gpurun_out/r04_c*.call 2>/dev/null | wc -l)
id=$(printf "r04_c%03d" $((n + 1)))
out="gpurun_out/$id.call"
/usr/local/graft/bin/gpurun "$@" > "$out" 2>&1
rc=$?
status=$(grep -o 'status=[a-z_]* rc=[-0-9]* charged=[0-9.]*s' "$out" | head -1)
left=$(grep -o 'GPU-minutes left this round: [0-9.]*' "$out" | tail -1 | grep -o '[0-9.]*$')
cmd="${@: -1}"
ledger=profiles/r04_gpurun_ledger.md
if [ ! -f "$ledger" ]; then
  printf '# Round 4 -- every gpurun call of the round (full client output: gpurun_out/<id>.call)\n\n| id | time (UTC) | result | GPU-min left | command |\n|---|---|---|---|---|\n' > "$ledger"
fi
printf '| %s | %s | %s exit %s | %s | `%s` |\n' "$id" "$(date -u +%H:%M:%S)" "${status:-no status line}" "$rc" "${left:-?}" "$(echo "$cmd" | tr '\n|' ' /' | cut -c1-400)" >> "$ledger"
tail -n 60 "$out"
exit $rc
