import os, sys
sys.path.insert(0, "/root/repo")
import torch
torch.cuda.init(); x = torch.zeros(4, device="cuda"); torch.cuda.synchronize()
print("=== torch up", flush=True); sys.stderr.write("=== MARK torch up\n")
import pos_evolution_amd as pea
e = pea.Engine(device=0)
sys.stderr.write("=== MARK engine A created\n")
e2 = pea.Engine(device=0)
sys.stderr.write("=== MARK engine B created\n")
e.close(); sys.stderr.write("=== MARK engine A closed\n")
e3 = pea.Engine(device=0)
sys.stderr.write("=== MARK engine C created\n")
