"""CPU oracle for the marginalized graph kernel -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this package; graphdot_amd/ never does."""
