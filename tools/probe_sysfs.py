import glob, os
for card in sorted(glob.glob('/sys/class/drm/card[0-9]*')):
    if '-' in os.path.basename(card): continue
    dev = os.path.realpath(card + '/device')
    print(card, '->', dev)
    for hw in glob.glob(card + '/device/hwmon/hwmon*'):
        for f in sorted(os.listdir(hw)):
            p = os.path.join(hw, f)
            if os.path.isfile(p) and any(f.startswith(x) for x in ('power', 'freq', 'temp', 'name')):
                try:
                    print('   ', f, open(p).read().strip()[:60])
                except Exception as e:
                    print('   ', f, 'ERR', e)
    for f in ('pp_dpm_sclk', 'pp_dpm_mclk', 'gpu_busy_percent', 'mem_busy_percent', 'current_link_speed'):
        p = card + '/device/' + f
        try:
            print('  ', f, open(p).read().strip().replace('\n', ' | ')[:200])
        except Exception as e:
            print('  ', f, 'ERR', e)
    p = card + '/device/gpu_metrics'
    try:
        b = open(p, 'rb').read()
        print('   gpu_metrics bytes', len(b), b[:4].hex())
    except Exception as e:
        print('   gpu_metrics ERR', e)
import torch
pr = torch.cuda.get_device_properties(0)
print([a for a in dir(pr) if 'pci' in a.lower()], getattr(pr, 'pci_bus_id', None), getattr(pr, 'pci_device_id', None), getattr(pr, 'pci_domain_id', None))
