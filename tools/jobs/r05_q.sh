cd /root/repo
python tools/host_lead.py 2>&1 | head -60
