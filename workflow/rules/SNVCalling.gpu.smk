### MI355X replacement of LongSom's workflow/rules/SNVCalling.smk (same rule names' OUTPUT files).
#
# One fused rule produces every file the reference chain
#   SplitBam -> BaseCellCounter (x2) -> MergeCounts -> BaseCellCalling_step1 -> _step2 -> _step3
# leaves behind (except the per-cell-type BAMs, which only BaseCellCounter consumed; rule SplitBam_gpu
# below writes them on request, e.g. for the fusion branch).  `include: "rules/SNVCalling.gpu.smk"`
# instead of "rules/SNVCalling.smk" in workflow/Snakefile; config keys are unchanged.

GPU_SCRIPTS = str(workflow.basedir) + "/scripts_gpu"

rule SNVCalling_gpu:
    input:
        bam=f"{INPUT}/bam/{{id}}.bam",
        bai=f"{INPUT}/bam/{{id}}.bam.bai",
        barcodes="CellTypeReannotation/ReannotatedCellTypes/{id}.tsv" if REANNO else "Barcodes/{id}.tsv",
        ref=str(workflow.basedir)+config['Reference']['genome'],
        pon_LR="PoN/PoN/PoN_LR.tsv" if PON else [],
        pon_SR=str(workflow.basedir)+config['Reference']['PoN_SR'],
        RNA_editing=str(workflow.basedir)+config['Reference']['RNA_editing'],
    output:
        report="SNVCalling/SplitBam/{id}.report.txt",
        counts=expand("SNVCalling/BaseCellCounter/{{id}}/{{id}}.{celltype}.tsv", celltype=['Cancer','Non-Cancer']),
        merged="SNVCalling/MergeCounts/{id}.BaseCellCounts.AllCellTypes.tsv",
        step1="SNVCalling/BaseCellCalling/{id}.calling.step1.tsv",
        step2="SNVCalling/BaseCellCalling/{id}.calling.step2.tsv",
        step3="SNVCalling/BaseCellCalling/{id}.calling.step3.tsv",
    params:
        script=GPU_SCRIPTS+"/SNVCalling/longsom_gpu_snv.py",
        c=config['SNVCalling']['BaseCellCalling'],
        mapq=config['SNVCalling']['BaseCellCounter']['min_mapping_quality'],
        gnomAD_db=str(workflow.basedir)+config['Reference']['gnomAD_db'],
        # Run.reference_gz_compat: True reproduces the reference's reading of the .gz position sets (opened as text, the
        # decode error is swallowed, the sets are EMPTY: SURVEY quirk Q1); default False = the sets are read
        gz_compat="--reference_gz_compat" if config['Run'].get('reference_gz_compat', False) else "",
        # Run.htslib_legacy_del_merge: True counts CIGAR 1D2D's first deleted column as 'D' (pysam over htslib <= 1.10); default: htslib >= 1.11
        htslib="--htslib_legacy_del_merge" if config['Run'].get('htslib_legacy_del_merge', False) else "",
        # Run.allow_missing_gnomad: True runs step 2 without its gnomAD filter when the database cannot be read (default: the rule fails, as the reference's gnomAD_DB() does)
        no_gnomad="--allow_missing_gnomad" if config['Run'].get('allow_missing_gnomad', False) else "",
        # GPUs of this node used by the rule: one rank per GPU, genomic regions sharded over the ranks
        launcher=lambda wc, resources: "python" if resources.gpu == 1 else f"python -m torch.distributed.run --standalone --local-addr 127.0.0.1 --nnodes=1 --nproc-per-node {resources.gpu}",
    resources:
        gpu=config['Run'].get('gpus', 1)
    log:
        "logs/SNVCalling_gpu/{id}.log",
    benchmark:
        "benchmarks/SNVCalling_gpu/{id}.benchmark.txt"
    shell:
        r"""
        {params.launcher} {params.script} \
        --bam {input.bam} --meta {input.barcodes} --ref {input.ref} --id {wildcards.id} --outdir SNVCalling \
        --editing {input.RNA_editing} --pon_SR {input.pon_SR} --pon_LR {input.pon_LR} --gnomAD_db {params.gnomAD_db} {params.gz_compat} {params.htslib} {params.no_gnomad} \
        --min_mapping_quality {params.mapq} \
        --min_cell_types {params.c[Min_cell_types]} --min_distance {params.c[min_distance]} \
        --max_gnomad_vaf {params.c[max_gnomAD_VAF]} --delta_vaf {params.c[deltaVAF]} --delta_mcf {params.c[deltaMCF]} \
        --min_ac_reads {params.c[min_ac_reads]} --min_ac_cells {params.c[min_ac_cells]} --clust_dist {params.c[clust_dist]} \
        --alpha1 {params.c[alpha1]} --beta1 {params.c[beta1]} --alpha2 {params.c[alpha2]} --beta2 {params.c[beta2]} > {log}
        """

rule SplitBam_gpu:
    input:
        bam=f"{INPUT}/bam/{{id}}.bam",
        barcodes="CellTypeReannotation/ReannotatedCellTypes/{id}.tsv" if REANNO else "Barcodes/{id}.tsv",
    output:
        expand("SNVCalling/SplitBam/{{id}}.{celltype}.bam", celltype=['Cancer','Non-Cancer'])
    params:
        script=GPU_SCRIPTS+"/PreProcessing/SplitBamCellTypes.py",
        mapq=config['SNVCalling']['BaseCellCounter']['min_mapping_quality'],
    shell:
        "python {params.script} --bam {input.bam} --meta {input.barcodes} --id {wildcards.id} --outdir SNVCalling/SplitBam --min_MQ {params.mapq}"
