import sys, os
sys.path.insert(0, os.getcwd())
order = sys.argv[1]
def maps():
    return sorted(set(l.split()[-1] for l in open('/proc/self/maps') if 'amdhip' in l or 'libhsa' in l or 'hiprtc' in l))
if order == "tsdr_first":
    from tempest_loader import load_package
    t = load_package(); c = t.Context(0); print("ctx ok", c.device_info()["name"]); print(maps())
    import torch; print("torch avail", torch.cuda.is_available(), torch.cuda.device_count()); print(maps())
else:
    import torch; print("torch avail", torch.cuda.is_available()); x = torch.zeros(4, device="cuda"); print(maps())
    from tempest_loader import load_package
    t = load_package(); c = t.Context(0); print("ctx ok", c.device_info()["name"]); print(maps())
