"""TEST INFRASTRUCTURE: CPU restatement of the reference's hot path and the recipe that pins
it to the reference. Imported only by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by slam_amd/."""
