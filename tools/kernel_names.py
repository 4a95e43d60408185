"""Short, UNIQUE names for the kernels of a rocprofv3 run.  Our own kernels are `lsg::name`; rocprim's arrive as kilobyte-long
trampoline_kernel<...> instantiations that differ only deep inside the template arguments (a truncated name collapses the three
kernels of the segmented sort - large, medium and small segments - into one key): they are named by their algorithm, the lambda
that tells the variants apart, and the key / value types."""
import hashlib
import re


def short_kernel_name(full: str) -> str:
    n = full.replace("void ", "")
    if "rocprim" not in n:
        return n.split("(")[0]
    algo = re.search(r"wrapped_(\w+?)_config", n)
    name = "rocprim:" + (algo.group(1) if algo else n.split("<")[0].split("::")[-1])
    if algo and algo.group(1) == "segmented_radix_sort":
        lam = re.findall(r"\{lambda\(auto:1\)#(\d)\}\)?$", n.strip())
        name += {"1": ":large_segments", "2": ":medium_segments", "3": ":small_segments"}.get(lam[0] if lam else "", "")
    elif "scan" in name:
        t = re.search(r"wrapped_scan_config<[^,]*,\s*([\w ]+)>", n)
        name += ":" + (t.group(1).replace(" ", "_") if t else "")
        if "max" in n.lower()[:1500] and "Max" in n:
            name += ":max"
    elif "init_lookback_scan_state" in n:
        name = "rocprim:init_lookback_scan_state"
    return name + "#" + hashlib.sha1(full.encode()).hexdigest()[:6]
