### MI355X replacement of the counting/calling half of LongSom's workflow/rules/PoN.smk (same OUTPUT files).
#
# The reference maps every normal (minimap2 / samtools / AddBarcodeTag: unchanged, keep those rules), then runs SplitBam ->
# BaseCellCounter -> MergeCounts -> BaseCellCalling.step1 per normal and aggregates the step-1 tables with PoN.py (awk | sort |
# datamash).  Option A (fused): rule PoN_gpu does all of that for all normals in one process per GPU; the per-normal tables are
# written only when config['PoN'].get('tables', True).  Option B: keep the reference's per-normal rules with the drop-in
# scripts of scripts_gpu/SNVCalling (see SNVCalling.gpu.smk) and take only rule PoN_files_gpu, the datamash-free PoN.py.
# BetaBinEstimation.py (VGAM through rpy2) is not replaced: its BetaBinEstimates.txt is an input here.

GPU_SCRIPTS = str(workflow.basedir) + "/scripts_gpu"

rule NormalsTable_gpu:
    input:
        bam=expand(f"{DATA}/bam/{{norm}}.bam", norm=NORMS),
        barcodes=expand(f"{DATA}/ctypes/{{norm}}.txt", norm=NORMS),
    output:
        temp(f"{OUTDIR}/PoN/normals.tsv")
    run:
        with open(output[0], "w") as out:
            for n, b, c in zip(NORMS, input.bam, input.barcodes):
                out.write("%s\t%s\t%s\n" % (n, b, c))

rule PoN_gpu:
    input:
        normals=f"{OUTDIR}/PoN/normals.tsv",
        bb=f"{OUTDIR}/PoN/PoN/BetaBinEstimates.txt",
    output:
        f"{OUTDIR}/PoN/PoN/PoN_LR.tsv"
    params:
        script=GPU_SCRIPTS+"/PoN/longsom_gpu_pon.py",
        hg38=config['Global']['genome'],
        mapq=config['SComatic']['BaseCellCounter']['min_mapping_quality'],
        alpha1=lambda w, input: get_BetaBinEstimates(input.bb, 'alpha1'),
        beta1=lambda w, input: get_BetaBinEstimates(input.bb, 'beta1'),
        alpha2=lambda w, input: get_BetaBinEstimates(input.bb, 'alpha2'),
        beta2=lambda w, input: get_BetaBinEstimates(input.bb, 'beta2'),
        p=config['PoN'],
        tables="" if config['PoN'].get('tables', True) else "--no_tables",
        # GPUs of this node used by the rule: the normals are spread over one rank per GPU
        launcher=lambda wc, resources: "python" if resources.gpu == 1 else f"python -m torch.distributed.run --standalone --local-addr 127.0.0.1 --nnodes=1 --nproc-per-node {resources.gpu}",
    resources:
        gpu=config.get('Run', {}).get('gpus', 1)
    shell:
        "{params.launcher} {params.script} --normals {input.normals} --ref {params.hg38} --outdir {OUTDIR} --min_mq {params.mapq} "
        "--alpha1 {params.alpha1} --beta1 {params.beta1} --alpha2 {params.alpha2} --beta2 {params.beta2} "
        "--min_ac_cells {params.p[min_ac_cells]} --min_ac_reads {params.p[min_ac_reads]} --min_cells {params.p[min_cells]} "
        "--min_cell_types {params.p[min_cell_types]} --min_samples 1 --rm_prefix No {params.tables}"

rule PoN_files_gpu:
    input:
        f"{OUTDIR}/PoN/BaseCellCalling/BaseCellCalling_files.txt"
    output:
        f"{OUTDIR}/PoN/PoN/PoN_LR.files.tsv"
    params:
        script=GPU_SCRIPTS+"/PoN/PoN.py",
    shell:
        "python {params.script} --in_tsv {input} --out_file {output} --min_samples 1 --rm_prefix No"
