bash scripts/exp/run_s8_4.sh 6 "renumbered or orientation or topological or tc_ or diamond or golden or short_rows or planted or clique or sorted or sort_neighbors or support"
bash scripts/exp/run_s8_5.sh "motif3" | grep -i "kst_\|orient\|relabel\|sorted\|task_rows\|edesc\|cb_task"
